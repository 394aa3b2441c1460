"""GPU parity tests (run with ``-m gpu`` on an MI355X), through the C ABI of libnlc_hip.so via the drop-in Python mirror:
MPPIDelay.command (a1-a5, a10-a12) against the reference fixtures and the oracle: fused / generic paths, literal closures, options, cost callables, the plain C client.  Helpers and tolerances: tests/gpu_common.py.
"""

import glob
import os

import numpy as np
import pytest
import torch

from gpu_common import *  # noqa: F401,F403
from gpu_common import GOLD, TOL, T64, load_sd, build_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g1_*.npz"))))
def test_mppi_oracle_dynamics_vs_reference_golden(nlc, path):
    """G1: whole command() with oracle dynamics + env cost vs the REAL reference MPPIDelay/oracle/env code."""
    g = np.load(path)
    env = "oderl-" + os.path.basename(path).split("_")[2]
    K, T, nx, nu, A, delay = int(g["K"]), int(g["T"]), int(g["nx"]), int(g["nu"]), float(g["A"]), int(g["delay"])

    def make(U0):
        return nlc.MPPIDelay(
            nlc.OracleDynamics(env, ts=0.05, delay=delay), nlc.EnvCost(env), nx, nlc.noise_sigma(nu),
            num_samples=K, horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A),
            u_scale=A, U_init=U0,
        )

    check_command_steps(nlc, g, make)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_mppi_nl_dynamics_vs_reference_golden(nlc, env, encoder_mode):
    """G3: command() with Neural-Laplace dynamics vs reference MPPIDelay + reference model."""
    g = np.load(f"{GOLD}/g3_nl_{env}.npz")
    model = build_model(nlc, load_sd(g))
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["d"]), int(g["nu"]), float(g["A"])

    def make(U0):
        return nlc.MPPIDelay(
            nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-" + env), d, nlc.noise_sigma(nu),
            num_samples=K, horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A),
            u_scale=A, U_init=U0,
        )

    check_command_steps(nlc, g, make)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_literal_nl_closures_are_recognised_and_plan_fused_g3(nlc, env):
    """an UNMODIFIED harness -- dynamics / running_cost are the literal closures of mppi_with_model.py
    (a local function over model + ts_pred, a local function over env) -- gets the fused planner: the constructor finds the
    model, the constant prediction time and the env inside the closures, the first command() verifies the candidates
    against the closures on a probe, and from then on `mppi.fused is True`.  Results: G3 (reference MPPIDelay + reference
    model) at the fused path's tolerance; the caller's torch RNG stream is not touched by the probe."""
    g = np.load(f"{GOLD}/g3_nl_{env}.npz")
    model = build_model(nlc, load_sd(g))
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["d"]), int(g["nu"]), float(g["A"])
    ts_pred = torch.tensor(0.05, device="cuda", dtype=torch.double).view(1, 1).repeat(K, 1)
    made = []

    def make(U0):
        dyn, cost = _literal_harness_closures("oderl-" + env, model=model, ts_pred=ts_pred)
        p = nlc.MPPIDelay(dyn, cost, d, nlc.noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0)
        assert not p.fused and p._candidate is not None  # recognised, not yet verified
        made.append(p)
        return p

    torch.manual_seed(123)
    before = torch.random.get_rng_state()
    with torch.no_grad():
        check_command_steps(nlc, g, make)
    assert torch.equal(before, torch.random.get_rng_state()), "the probe must not consume the caller's RNG stream"
    p = made[0]
    assert p.fused is True and p.recognised is True and isinstance(p.F, nlc.NLDynamics) and isinstance(p.running_cost, nlc.EnvCost)
    p.ctx.profile_reset()
    p.ctx.profile(True)
    p.command(g["s1_state"], T64(g["s1_action_buffer"]))
    p.ctx.profile(False)
    assert any(k in p.ctx.profile_read() for k in ("nl_plan_fused_kernel", "nl_rollout_kernel")), p.ctx.profile_read()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "g1_*_d[02].npz"))))
def test_literal_oracle_partial_is_recognised_and_plans_fused_g1(nlc, path):
    """The harness's oracle branch: functools.partial(<env>_dynamics_dt_delay, ts=ts_pred, delay=, friction=) -> fused
    oracle rollout, G1 parity (real reference MPPIDelay / oracle.py / env rewards)."""
    g = np.load(path)
    env = "oderl-" + os.path.basename(path).split("_")[2]
    K, T, nx, nu, A, delay = int(g["K"]), int(g["T"]), int(g["nx"]), int(g["nu"]), float(g["A"]), int(g["delay"])
    ts_pred = torch.tensor(0.05, dtype=torch.double).view(1, 1).repeat(K, 1)
    made = []

    def make(U0):
        dyn, cost = _literal_harness_closures(env, ts_pred=ts_pred, delay=delay)
        p = nlc.MPPIDelay(dyn, cost, nx, nlc.noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0)
        made.append(p)
        return p

    check_command_steps(nlc, g, make)
    assert made[0].fused is True and isinstance(made[0].F, nlc.OracleDynamics)


def test_closure_that_differs_from_its_candidate_stays_generic(nlc):
    """Two gates.  Structure: a closure over a model + constant ts_pred that ALSO clamps the state references a
    name the harness closure does not (`clamp`) and carries constants of its own: it is never a candidate; nor is state MINUS
    the model's prediction (round 5, ADVICE r4: no name, no constant gives it away, its one operator does).  Probe: a closure
    with the harness closure's names, constants AND operations that computes something else (the prediction added to itself,
    the state dropped) is a candidate, fails the probe and keeps the generic path with the closure's own semantics."""
    g = np.load(f"{GOLD}/g3_nl_cartpole.npz")
    model = build_model(nlc, load_sd(g))
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["d"]), int(g["nu"]), float(g["A"])
    ts_pred = torch.full((K, 1), 0.05, dtype=torch.double, device="cuda")

    def clamped(state, perturbed_action):
        return (state + model(state, perturbed_action, ts_pred)).clamp(-0.5, 0.5)

    def minus(state, perturbed_action):
        state_diff_pred = model(state, perturbed_action, ts_pred)
        return state - state_diff_pred

    def doubled(state, perturbed_action):
        state_diff_pred = model(state, perturbed_action, ts_pred)
        return state_diff_pred + state_diff_pred

    cost = nlc.EnvCost("oderl-cartpole")
    with torch.no_grad():
        for dynamics, is_candidate in ((clamped, False), (minus, False), (doubled, True)):
            p = nlc.MPPIDelay(dynamics, cost, d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0, u_min=torch.tensor(-A),
                              u_max=torch.tensor(A), u_scale=A, U_init=torch.zeros(T, nu, dtype=torch.float64))
            assert (p._candidate is not None) == is_candidate, dynamics.__name__
            p.command(g["s0_state"], T64(g["s0_action_buffer"]))
            assert p.fused is False and p.recognised is False and p.F is dynamics, dynamics.__name__
            if dynamics is clamped:
                assert float(p.states.abs().max()) <= 0.5


@pytest.mark.parametrize("env", ["cartpole", "acrobot"])
def test_mppi_generic_callables_match_fused(nlc, env):
    """The external-callable path (reference contract: arbitrary closures) equals the fused path."""
    g = np.load(f"{GOLD}/g3_nl_{env}.npz")
    model = build_model(nlc, load_sd(g))
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["d"]), int(g["nu"]), float(g["A"])
    dyn, cost = nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-" + env)

    def make(U0):
        with torch.no_grad():
            return nlc.MPPIDelay(
                lambda s, a: dyn(s, a), lambda s, u: cost(s, u), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0,
            )

    with torch.no_grad():
        check_command_steps(nlc, g, make)


def test_mppi_seeded_torch_noise_matches_oracle_class(nlc):
    """Identical seeds => identical noise stream as the reference-style oracle (ctor draw + command draws)."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi

    env, K, T, A = "oderl-pendulum", 96, 6, 2.0
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    torch.manual_seed(42)
    ref = omppi.MPPIOracle(lambda s, w: oenvs.ORACLE_DYNAMICS[env](s, w, ts, 1), oenvs.RUNNING_COST[env], 3,
                           nlc.noise_sigma(1), K, T, 1.0, torch.tensor(-A), torch.tensor(A), A)
    state = oenvs.initial_state(env)
    ab = torch.zeros(4, 1, dtype=torch.float64)
    ref_actions = [ref.command(state, ab).clone() for _ in range(3)]
    torch.manual_seed(42)
    mine = nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, 1), nlc.EnvCost(env), 3, nlc.noise_sigma(1), K, T, "cpu",
                         lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)
    for a_ref in ref_actions:
        np.testing.assert_allclose(mine.command(state, ab).numpy(), a_ref.numpy(), **TOL)
    mine.reset()
    assert mine.U.shape == (T, 1)


def test_mppi_options_null_action_abs_cost_per_sample_state(nlc):
    from oracle import envs as oenvs
    from oracle import mppi as omppi

    env, K, T, A, nu, nx = "oderl-acrobot", 80, 5, 5.0, 2, 6  # K not a multiple of 64
    torch.manual_seed(1)
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    U0 = torch.randn(T, nu, dtype=torch.float64) * 0.3
    raw = torch.randn(K, T, nu, dtype=torch.float64)
    states0 = torch.stack([oenvs.initial_state(env, s) for s in range(K)])
    ab = torch.randn(4, nu, dtype=torch.float64)
    sig = nlc.noise_sigma(nu)
    mine = nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, 2), nlc.EnvCost(env), nx, sig, K, T, "cpu", lambda_=0.7,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=torch.roll(U0, 1, 0),
                         u_per_command=2, sample_null_action=True, noise_abs_cost=True)
    # U_init rolled by +1 so that command()'s roll(-1) restores rows 0..T-2 of U0 ... except the last row,
    # which becomes u_init = 0: give the oracle the same starting point
    U_start = torch.roll(U0, 1, 0)
    ref = omppi.mppi_command(U_start.clone(), states0, ab, raw.clone(),
                             lambda s, w: oenvs.ORACLE_DYNAMICS[env](s, w, ts, 2), oenvs.RUNNING_COST[env], nx,
                             torch.inverse(sig), 0.7, A, torch.tensor(-A), torch.tensor(A), sample_null_action=True,
                             noise_abs_cost=True, u_per_command=2)
    mine.noise_dist = type("Replay", (), {"sample": staticmethod(lambda shape: raw)})()
    act = mine.command(states0, ab)
    assert act.shape == (2, nu)
    np.testing.assert_allclose(act.numpy(), ref["action"].numpy(), **TOL)
    np.testing.assert_allclose(mine.cost_total.numpy(), ref["cost_total"].numpy(), **TOL)
    np.testing.assert_allclose(mine.omega.numpy(), ref["omega"].numpy(), **TOL)
    assert torch.all(mine.perturbed_action[-1] == 0)


def test_mppi_philox_noise_statistics_and_shard_invariance(nlc):
    """Device RNG: N(0, Sigma) moments, determinism in (seed, command index), and K-shard invariance."""
    env, K, T, A, nu = "oderl-acrobot", 4096, 10, 5.0, 2
    sig = nlc.noise_sigma(nu)

    def planner(**kw):
        return nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, 0), nlc.EnvCost(env), 6, sig, K, T, "cuda", lambda_=1.0,
                             u_scale=A, U_init=torch.zeros(T, nu, dtype=torch.float64), noise_rng="philox", seed=7, **kw)

    st, ab = nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
    a = planner()
    a.command(st, ab)
    eps = a.noise.cpu().reshape(-1, nu)  # no bounds => noise is the raw draw
    assert abs(float(eps.mean())) < 0.02
    cov = (eps.T @ eps) / eps.shape[0]
    np.testing.assert_allclose(cov.numpy(), sig.numpy(), atol=0.03)
    kurt = float((eps[:, 0] ** 4).mean())
    assert abs(kurt - 3.0) < 0.15
    b = planner()
    b.command(st, ab)
    assert torch.equal(a.noise, b.noise)
    # a shard configured with k_offset draws exactly its slice of the global stream
    half = planner()
    half.K_local, half.k_offset = K // 2, K // 2
    half.command(st, ab)
    assert torch.equal(half.noise, a.noise[K // 2 :])
    # the device-generated noise fed back through the CPU oracle reproduces the device result
    from oracle import envs as oenvs
    from oracle import mppi as omppi

    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    ref = omppi.mppi_command(torch.zeros(T, nu, dtype=torch.float64), st, ab, a.noise.cpu().clone(),
                             lambda s, w: oenvs.ORACLE_DYNAMICS[env](s, w, ts, 0), oenvs.RUNNING_COST[env], 6,
                             torch.inverse(sig), 1.0, A)
    np.testing.assert_allclose(a.cost_total.cpu().numpy(), ref["cost_total"].numpy(), **TOL)
    np.testing.assert_allclose(a.U.cpu().numpy(), ref["U"].numpy(), **TOL)


@pytest.mark.parametrize("env", ["pendulum", "acrobot"])
def test_rollout_samples_vs_reference_golden(nlc, env):
    """G13: rollout_samples = 3 with a rollout_var_cost (mppi_delay.py:291-292, 310) vs the REAL reference planner, on
    the fused oracle-dynamics rollout and on the generic callable path."""
    from oracle import envs as oenvs

    g = np.load(f"{GOLD}/g13_rollout_samples_{env}.npz")
    name = "oderl-" + env
    d, nu, K, T, A, delay = int(g["nx"]), int(g["nu"]), int(g["K"]), int(g["T"]), float(g["A"]), int(g["delay"])
    kw = dict(rollout_samples=int(g["M"]), rollout_var_cost=float(g["var_cost"]), rollout_var_discount=float(g["var_discount"]))
    ts = torch.full((K, 1), 0.05, dtype=torch.float64, device="cuda")

    def make(U0, fused=True):
        if fused:
            dyn, cost = nlc.OracleDynamics(name, 0.05, delay), nlc.EnvCost(name)
        else:
            dyn = lambda s, w: oenvs.ORACLE_DYNAMICS[name](s, w, ts, delay)  # noqa: E731  (torch ops on the device)
            cost = lambda s, u: nlc.EnvCost(name)(s, u)  # noqa: E731
        p = nlc.MPPIDelay(dyn, cost, d, nlc.noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0, **kw)
        assert p.fused == fused
        return p

    check_command_steps(nlc, g, make)
    check_command_steps(nlc, g, lambda U0: make(U0, fused=False))


def test_error_paths_raise(nlc):
    from neurallaplacecontrol_amd import _lib

    with pytest.raises(NotImplementedError):
        nlc.ilt_reconstruct(torch.zeros(2, 1, 17).double().cuda(), torch.zeros(2, 1, 17).double().cuda(),
                            torch.ones(2).double().cuda(), "cme")
    for bad_terms in (20, 35, 1):  # de Hoog needs an odd number of terms, 3 .. 33 (2M + 1)
        with pytest.raises(_lib.NlcError):
            z = torch.zeros(2, 1, bad_terms).double().cuda()
            nlc.ilt_reconstruct(z, z.clone(), torch.ones(2).double().cuda(), "dehoog")
    with pytest.raises(ValueError):
        nlc.ilt_reconstruct(torch.zeros(2, 1, 17).double().cuda(), torch.zeros(2, 2, 17).double().cuda(),
                            torch.ones(2).double().cuda())
    with pytest.raises(_lib.NlcError):  # nx does not match the env
        m = nlc.MPPIDelay(nlc.OracleDynamics("oderl-pendulum"), nlc.EnvCost("oderl-pendulum"), 5, nlc.noise_sigma(1), 8, 4)
        m.command(torch.zeros(5).double(), torch.zeros(4, 1).double())
    with pytest.raises(_lib.NlcError):  # delay beyond the action buffer
        m = nlc.MPPIDelay(nlc.OracleDynamics("oderl-pendulum", delay=4), nlc.EnvCost("oderl-pendulum"), 3,
                          nlc.noise_sigma(1), 8, 4)
        m.command(torch.zeros(3).double(), torch.zeros(4, 1).double())


def test_collector_variant_encode_obs_time_with_oracle_dynamics(nlc):
    """Dataset-collector call pattern (mppi_dataset_collector.py:166-180,249): encode_obs_time=True, action_buffer
    carries an extra time-stamp column, oracle dynamics ignore it -> same result as the plain call."""
    env, K, T, A = "oderl-cartpole", 128, 8, 3.0
    torch.manual_seed(2)
    raw = torch.randn(K, T, 1, dtype=torch.float64)
    U0 = torch.randn(T, 1, dtype=torch.float64) * 0.2
    st = nlc.initial_state(env)
    ab = torch.randn(4, 1, dtype=torch.float64)
    ab_t = torch.cat((ab, torch.tensor([[0.15], [0.10], [0.05], [0.0]], dtype=torch.float64)), dim=1)
    acts = []
    for enc, buf in ((False, ab), (True, ab_t)):
        m = nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, 1), nlc.EnvCost(env), 5, nlc.noise_sigma(1), K, T, "cpu",
                          lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone(),
                          encode_obs_time=enc)
        assert m.fused
        m.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
        acts.append(m.command(st, buf))
    assert torch.equal(acts[0], acts[1])


def test_get_rollouts_open_loop_replay(nlc):
    """MPPIDelay.get_rollouts (reference :358-381): open-loop replay of U through the dynamics callable."""
    from oracle import nl_model as onl

    env, T, A = "oderl-cartpole", 6, 3.0
    st = onl.ENV_STATS[env]
    sd = onl.make_synthetic_state_dict(0, 5, 1, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    U0 = torch.linspace(-0.5, 0.5, T, dtype=torch.float64).view(T, 1)
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), 5, nlc.noise_sigma(1), 64, T, "cpu",
                         u_scale=A, U_init=U0.clone())
    x0 = nlc.initial_state(env)
    with torch.no_grad():
        got = mppi.get_rollouts(x0)
    ts = torch.full((1, 1), 0.05, dtype=torch.float64)
    x, ref = x0.view(1, -1), []
    for t in range(T):
        x = x + onl.nl_forward(sd, x, (A * U0[t]).view(1, 1, 1), ts, S=17).view(1, -1)
        ref.append(x)
    np.testing.assert_allclose(got.numpy(), torch.stack(ref, dim=1).numpy(), **TOL)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_collector_variant_vs_reference_golden(nlc, env):
    """G5a: reference MPPIDelay(encode_obs_time=True) + reference oracle dynamics with the (B, nu+1) action buffer
    of the dataset collector; the fused path reproduces it and leaves the caller's buffer untouched."""
    g = np.load(f"{GOLD}/g5_collector_{env}.npz")
    K, T, nx, nu, A, delay = int(g["K"]), int(g["T"]), int(g["nx"]), int(g["nu"]), float(g["A"]), int(g["delay"])

    def make(U0):
        m = nlc.MPPIDelay(
            nlc.OracleDynamics("oderl-" + env, ts=0.05, delay=delay), nlc.EnvCost("oderl-" + env), nx, nlc.noise_sigma(nu),
            num_samples=K, horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A),
            u_scale=A, U_init=U0, encode_obs_time=True, dt=float(g["dt"]),
        )
        assert m.fused
        return m

    check_command_steps(nlc, g, make)
    ab = T64(g["s0_action_buffer"])
    keep = ab.clone()
    make(T64(g["s0_U_before"])).command(g["s0_state"], ab)
    assert torch.equal(ab, keep)


@pytest.mark.parametrize("variant", ["constraint", "goal", "goal_flipped"])
@pytest.mark.parametrize("dyn_name", ["nl", "oracle"])
def test_cost_callables_with_fused_dynamics_vs_reference_golden(nlc, variant, dyn_name):
    """The harness running_cost's state_constraint / change_goal branches (mppi_with_model.py:146-162) and a
    terminal_state_cost are arbitrary callables: the rollout still runs in the fused kernel (cost_external), the
    callables on the stored device states.  Golden: the real planner + the real cartpole env class."""
    from oracle import envs as oenvs

    g = np.load(f"{GOLD}/g8_cost_variants.npz")
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["d"]), int(g["nu"]), float(g["A"])
    kw = {"constraint": dict(state_constraint=True), "goal": dict(change_goal=True),
          "goal_flipped": dict(change_goal=True, change_goal_flipped=True)}[variant]
    cost = oenvs.cartpole_cost_variant(**kw)  # stands in for the harness closure (the env class is not on this box)
    terminal = (lambda states, actions: 0.5 * (states[..., -1, 0] ** 2).reshape(-1)) if variant == "goal" else None
    dyn = nlc.NLDynamics(build_model(nlc, load_sd(g)), 0.05) if dyn_name == "nl" else nlc.OracleDynamics("oderl-cartpole", 0.05, 1)
    calls = []

    def counted_cost(state, action):
        calls.append((tuple(state.shape), state.device.type))
        return cost(state, action)

    def make(U0):
        p = nlc.MPPIDelay(dyn, counted_cost, d, nlc.noise_sigma(nu), num_samples=K, horizon=T, device="cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0, terminal_state_cost=terminal)
        assert p.fused_dynamics and p.cost_external and not p.fused
        return p

    g2 = {k[len(f"{variant}_{dyn_name}_"):]: g[k] for k in g.files if k.startswith(f"{variant}_{dyn_name}_")}
    with torch.no_grad():
        check_command_steps(nlc, g2, make)
    assert calls and all(c == ((K, d), "cuda") for c in calls) and len(calls) == 2 * T


def test_cost_callables_path_equals_fused_envcost(nlc):
    """With the default cost written as a callable, the cost_external path gives the fused EnvCost result."""
    from oracle import envs as oenvs
    from oracle import nl_model as onl

    env, d, nu, A, K, T = "oderl-acrobot", 6, 2, 5.0, 200, 7
    st = onl.ENV_STATS[env]
    sd = onl.make_synthetic_state_dict(21, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    gen = torch.Generator().manual_seed(5)
    raw = torch.randn(K, T, nu, dtype=torch.float64, generator=gen)
    U0 = torch.randn(T, nu, dtype=torch.float64, generator=gen) * 0.2
    state, ab = _state(nlc, env, 4), torch.randn(4, nu, dtype=torch.float64, generator=gen)
    out = []
    for rc in (nlc.EnvCost(env), oenvs.RUNNING_COST[env]):
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), rc, d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone())
        p.noise_dist = _Replay(raw.clone())
        with torch.no_grad():
            act = p.command(state, ab)
        out.append((act, p.cost_total.clone(), p.states.clone(), p.U.clone()))
    assert torch.equal(out[0][2], out[1][2])  # same kernel, same states
    for a, b in zip(out[0], out[1]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-10, atol=1e-12)


def test_plain_c_client_of_the_abi_matches_the_python_mirror(nlc, tmp_path):
    """include/nlc.h is a real C boundary: tests/helpers/cabi_client.c (C99, gcc, HIP runtime C API for the device
    buffers, no Python, no torch) runs three planner commands; the Python mirror driving the same library with the same
    seed / command counters gives bit-identical numbers."""
    import subprocess

    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    libdir = os.path.join(repo, "neurallaplacecontrol_amd")
    exe = str(tmp_path / "cabi_client")
    subprocess.check_call(
        ["gcc", "-std=c99", os.path.join(repo, "tests", "helpers", "cabi_client.c"), "-I", os.path.join(repo, "include"),
         "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L" + libdir, "-lnlc_hip", "-L/opt/rocm/lib", "-lamdhip64",
         "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])
    out = subprocess.check_output([exe], timeout=300, env=dict(os.environ, NCCL_DEBUG="WARN")).decode()

    def numeric(ln):  # (RCCL may still print a banner to stdout at communicator creation)
        try:
            [float(x) for x in ln.split()]
            return bool(ln.split())
        except ValueError:
            return False

    lines = [ln for ln in out.strip().splitlines() if numeric(ln)]
    c_cmds = [[float(x) for x in ln.split()] for ln in lines[:3]]
    c_U = [float(x) for x in lines[3].split()]
    K, T, A = 512, 10, 3.0
    p = nlc.MPPIDelay(nlc.OracleDynamics("oderl-cartpole", 0.05, 2), nlc.EnvCost("oderl-cartpole"), 5, torch.tensor(1.0).double(),
                      K, T, "cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
                      U_init=torch.zeros(T, 1, dtype=torch.float64), noise_rng="philox", seed=17)
    state = torch.tensor([0.01, 0.0, -1.0, 0.02, 0.0], dtype=torch.float64)
    ab = torch.tensor([[0.5], [-0.25], [0.0], [1.0]], dtype=torch.float64)
    for cmd in range(3):  # (the client's third command gathers with the library's own one-rank RCCL communicator)
        act = p.command(state, ab)
        part = p._partials.cpu()
        assert [float(act[0]), float(p.cost_total[0]), float(part[0]), float(part[1])] == c_cmds[cmd]
    assert p.U.reshape(-1).tolist() == c_U
    # env step through the C client == through BatchedEnv
    c_env = [float(x) for x in lines[4].split()]
    e = nlc.BatchedEnv("oderl-cartpole", 2, dt=0.05, action_delay=1, action_buffer_size=3)
    e.set_state_(torch.tensor([[0.1, -0.2, 3.0, 0.5], [-0.3, 0.4, 2.5, -1.0]], dtype=torch.float64))
    e.action_buffer.copy_(torch.tensor([0.5, 1.0, -2.0, 0.25, -0.5, 1.5], dtype=torch.float64).view(2, 3, 1))
    obs, rew = e.step(torch.tensor([[2.0], [-1.0]], dtype=torch.float64))
    assert obs.cpu().reshape(-1).tolist() + rew.cpu().tolist() == c_env
    # ILT forward / backward through the C client == through the Python mirror's autograd
    # (one line per algorithm: forward kernels AND backward kernels of fourier, dehoog, fixed_tablot)
    N, D, S = 3, 2, 17
    i = torch.arange(N * D * S, dtype=torch.float64)
    w1 = (1 + torch.arange(N * D * S) % 3).double().cuda()
    w2 = (1 + torch.arange(N * D * S) % 5).double().cuda()
    gx = (1.0 + 0.5 * torch.arange(N * D, dtype=torch.float64)).view(N, D).cuda()
    for m, algo in enumerate(("fourier", "dehoog", "fixed_tablot")):
        c_ilt = [float(x) for x in lines[5 + m].split()]
        th = (3.0 * ((i * 37) % 101) / 101.0 - 1.5).view(N, D, S).cuda().requires_grad_()
        ph = (1.2 * ((i * 53) % 97) / 97.0 - 0.6).view(N, D, S).cuda().requires_grad_()
        x = nlc.ilt_reconstruct(th, ph, torch.tensor([0.1, 0.125, 0.3], dtype=torch.float64).cuda(), algo)
        gt, gp = torch.autograd.grad(x, (th, ph), gx)
        np.testing.assert_allclose(x.detach().cpu().reshape(-1).numpy(), c_ilt[: N * D], rtol=0, atol=0, err_msg=algo)
        np.testing.assert_allclose([float((gt.reshape(-1) * w1).sum()), float((gp.reshape(-1) * w2).sum())], c_ilt[N * D:],
                                   rtol=1e-13, err_msg=algo)


class _Prefixed:
    """View of the fixture keys that start with a prefix (G15 stores several cases in one file)."""

    def __init__(self, g, prefix):
        self.g, self.p = g, prefix

    def __getitem__(self, k):
        return self.g[self.p + k]


@pytest.mark.parametrize("case", ["o0_", "o2_", "n_"])
def test_state_dim_4_cartpole_without_trig_vs_reference_golden(nlc, case):
    """BASELINE's literal state_dim = 4 on the FUSED planner: CTCartpole(obs_trans=False), state [x, xdot, theta, thetadot]
    (env "oderl-cartpole-notrig": running cost ctcartpole.py:297-300, oracle dynamics oracle.py:38-44 / 80-86, NL dynamics at
    d = 4) against G15 -- the reference's MPPIDelay driving its own oracle function / its own NeuralLaplaceModel(state_dim=4)
    and the real env's reward methods."""
    g = np.load(os.path.join(GOLD, "g15_notrig_cartpole.npz"))
    env = "oderl-cartpole-notrig"
    K, Tn, nx, nu, A, S = int(g["K"]), int(g["T"]), int(g["nx"]), int(g["nu"]), float(g["A"]), int(g["S"])
    if case == "n_":
        model = build_model(nlc, load_sd(g), S=S)
        with torch.no_grad():
            fwd = model(T64(g["fwd_obs"]).cuda(), T64(g["fwd_window"]).cuda(), T64(g["fwd_ts"]).cuda()).cpu()
        np.testing.assert_allclose(fwd.numpy(), g["fwd_out"], **TOL)
        dyn = nlc.NLDynamics(model, 0.05)
    else:
        dyn = nlc.OracleDynamics(env, 0.05, int(case[1]))

    def make(U0):
        p = nlc.MPPIDelay(dyn, nlc.EnvCost(env), nx, nlc.noise_sigma(nu), K, Tn, "cpu", lambda_=1.0, u_min=torch.tensor(-A),
                          u_max=torch.tensor(A), u_scale=A, U_init=U0)
        assert p.fused
        return p

    with torch.no_grad():
        check_command_steps(nlc, _Prefixed(g, case), make)


def test_plain_c_client_walks_the_sharded_protocol_with_a_forced_give_up(nlc, tmp_path):
    """the protocol a C caller with its OWN collective implements, walked by a C99 program
    (tests/helpers/cabi_sharded_client.c): one population of 1024 samples as two shards on two ctxs, Neural-Laplace dynamics on
    the fused one-launch body, the collective = two device copies.  In command 0 shard 1's fused launch gives up
    (fused_test_drop_tile): nlc_mppi_finish returns NLC_AGAIN on BOTH ctxs, the client gathers the re-run's rows again and calls
    again.  Both ctxs return the same bits in every command, those are the unsharded Python planner's (same weights from the
    same LCG, same Philox seed / counters), only shard 1 leaves the fused body, and nlc_get_stat says so."""
    import subprocess

    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    libdir = os.path.join(repo, "neurallaplacecontrol_amd")
    exe = str(tmp_path / "cabi_sharded_client")
    subprocess.check_call(
        ["gcc", "-std=c99", os.path.join(repo, "tests", "helpers", "cabi_sharded_client.c"), "-I", os.path.join(repo, "include"),
         "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L" + libdir, "-lnlc_hip", "-L/opt/rocm/lib", "-lamdhip64",
         "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])
    lines = subprocess.check_output([exe], timeout=300).decode().strip().splitlines()
    nblob = int(lines[0])
    cmds = [ln.split() for ln in lines[1:4]]
    stats = [[float(x) for x in lines[4].split()], [float(x) for x in lines[6].split()]]
    U_c = [[float(x) for x in lines[5].split()], [float(x) for x in lines[7].split()]]
    # both ranks asked for the second gather in command 0 only, and returned the same bits every time
    assert [(c[2], c[3]) for c in cmds] == [("1", "1"), ("0", "0"), ("0", "0")], cmds
    assert all(c[0] == c[1] for c in cmds) and U_c[0] == U_c[1]
    # rank 0 still plans on the fused body (3) and lost nothing itself; rank 1 lost one launch and stays on the latency-split
    # body (2); both re-ran one command, command 0
    assert stats[0] == [3.0, 0.0, 1.0, 0.0] and stats[1] == [2.0, 1.0, 1.0, 0.0], stats
    # the same model in the Python mirror: weights from the same 64-bit LCG, blob order = state_dict order (include/nlc.h)
    d, nu, S, h, T, K, A = 5, 1, 17, 128, 12, 1024, 3.0
    x, vals, mask = 0x9E3779B97F4A7C15, [], (1 << 64) - 1
    for _ in range(nblob):
        x = (x * 6364136223846793005 + 1442695040888963407) & mask
        vals.append(((x >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 0.3)
    blob = torch.tensor(vals, dtype=torch.float64)
    blob[nblob - 2 * d * S + d * S:] += -3.0
    model = nlc.NeuralLaplaceModel(d, nu, d, hidden_units=h, s_recon_terms=S, ilt_algorithm="fourier", state_mean=np.zeros(d),
                                   state_std=np.array([2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048]),
                                   action_mean=np.array([0]), action_std=np.array([1.5]), normalize=True, normalize_time=True).double()
    order = ([f"action_encoder.gru.{n}_l{l}" for l in (0, 1) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
             + ["action_encoder.linear_out.weight", "action_encoder.linear_out.bias"]
             + [f"laplace_rep_func.linear_tanh_stack.{i}.{n}" for i in (0, 2, 4) for n in ("weight", "bias")])
    sd, at = model.state_dict(), 0
    for name in order:
        n = sd[name].numel()
        sd[name] = blob[at:at + n].view(sd[name].shape).clone()
        at += n
    assert at == nblob
    model.load_state_dict(sd)
    model = model.cuda()
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-cartpole"), d, torch.tensor(1.0).double(), K, T, "cpu",
                      lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A,
                      U_init=torch.zeros(T, nu, dtype=torch.float64), noise_rng="philox", seed=23)
    state = torch.tensor([0.01, 0.0, -1.0, 0.02, 0.0], dtype=torch.float64)
    ab = torch.tensor([[0.5], [-0.25], [0.0], [1.0]], dtype=torch.float64)
    for cmd in range(3):
        act = p.command(state, ab)
        # (two shards merged through the beta-shifted sums vs one population: the reductions associate differently)
        np.testing.assert_allclose(float(cmds[cmd][0]), float(act[0]), rtol=1e-10, atol=1e-12)
        ab = torch.roll(ab, -1, 0)
        ab[-1] = float(cmds[cmd][0])
    np.testing.assert_allclose(np.array(U_c[0]), p.U.reshape(-1).numpy(), rtol=1e-9, atol=1e-12)


def test_refresh_model_picks_up_a_write_through_data(nlc):
    """A write through `.data` moves no version counter (documented in _weights.py): the planner keeps planning with the weights
    it uploaded until `MPPIDelay.refresh_model()` (round 5) -- after which a command equals a fresh planner's."""
    from oracle import nl_model as onl

    env, K, T = "oderl-cartpole", 256, 8
    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(2, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)

    def make():
        return nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                             u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=9,
                             U_init=torch.zeros(T, nu, dtype=torch.float64))

    state, ab = nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
    p = make()
    a0 = p.command(state, ab).clone()
    with torch.no_grad():
        model.laplace_rep_func.linear_tanh_stack[4].weight.data.mul_(1.7)
    p.U = torch.zeros(T, nu, dtype=torch.float64)
    p._commands = 0  # the same Philox counters as the first command
    a_stale = p.command(state, ab).clone()
    assert torch.equal(a_stale, a0), "a .data write is invisible to the weights key: the planner still holds the old weights"
    p.refresh_model()
    p.U = torch.zeros(T, nu, dtype=torch.float64)
    p._commands = 0
    a_new = p.command(state, ab).clone()
    fresh = make().command(state, ab)
    assert torch.equal(a_new, fresh) and not torch.equal(a_new, a0)


def _toy_dynamics_and_cost(nx, nu, dtype):
    """A smooth nonlinear system with nu action dims and a delay window (the reference's contract: dynamics(state, window),
    running_cost(state, u), planners/mppi_delay.py:271-296) -- plain tensor ops, the same code for the oracle and the planner."""
    g = torch.Generator().manual_seed(5)
    Wx = (torch.randn(nx, nx, generator=g, dtype=torch.float64) * 0.3).to(dtype)
    Wu = (torch.randn(nu, nx, generator=g, dtype=torch.float64) * 0.5).to(dtype)

    def dynamics(state, window):
        u = 0.7 * window[:, -1, :] + 0.3 * window[:, 0, :]
        return state + 0.05 * (torch.tanh(state @ Wx.to(state.device)) + u @ Wu.to(state.device))

    def cost(state, u):
        return (state**2).sum(dim=1) + 0.01 * (u**2).sum(dim=1)

    return dynamics, cost


@pytest.mark.parametrize("dtype,nu,tol", [(torch.float64, 3, 1e-9), (torch.float64, 5, 1e-9), (torch.float32, 1, 1e-4), (torch.float32, 3, 1e-4)])
def test_planner_limits_are_paths_nu_above_two_and_float32(nlc, dtype, nu, tol):
    """the reference class takes any nu and dtype (planners/mppi_delay.py:115-135); the drop-in constructor
    must not throw where the HIP planner kernels are not built (nu > 2, a float32 noise_sigma): it plans with tensor ops on the
    GPU and says so once.  Seeded like the reference (ctor draw, command draws), against the oracle class."""
    from oracle import mppi as omppi

    nx, K, T, B, A = 4, 256, 12, 4, 2.0
    dyn, cost = _toy_dynamics_and_cost(nx, nu, dtype)
    sig = nlc.noise_sigma(nu).to(dtype) if nu > 1 else torch.tensor(1.0, dtype=dtype)
    state = torch.linspace(-0.5, 0.5, nx, dtype=dtype)
    ab = (torch.arange(B * nu, dtype=dtype).view(B, nu) * 0.05) - 0.1
    torch.manual_seed(11)
    ref = omppi.MPPIOracle(dyn, cost, nx, sig, K, T, 1.0, torch.tensor(-A, dtype=dtype), torch.tensor(A, dtype=dtype), A)
    ref_actions = [ref.command(state, ab).clone() for _ in range(3)]
    torch.manual_seed(11)
    with pytest.warns(UserWarning, match="PyTorch-ROCm tensor ops"):
        mine = nlc.MPPIDelay(dyn, cost, nx, sig, K, T, "cpu", lambda_=1.0, u_min=torch.tensor(-A, dtype=dtype),
                             u_max=torch.tensor(A, dtype=dtype), u_scale=A)
    assert mine.torch_path and mine.rollout_body is None
    for a_ref in ref_actions:
        a = mine.command(state, ab)
        assert a.dtype == dtype and a.shape == (nu,)
        np.testing.assert_allclose(a.double().numpy(), a_ref.double().numpy(), rtol=tol, atol=tol)
    assert mine.rollout_body == "callables-torch" and mine._states.is_cuda
    last = ref.last
    np.testing.assert_allclose(mine.cost_total.double().numpy(), last["cost_total"].double().numpy(), rtol=tol * 10, atol=tol * 10)
    np.testing.assert_allclose(mine.states.double().numpy(), last["states"].double().numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(mine.omega.double().numpy(), last["omega"].double().numpy(), rtol=tol * 10, atol=tol)
    np.testing.assert_allclose(mine.U.double().numpy(), ref.U.double().numpy(), rtol=tol, atol=tol)
    mine.reset()
    assert mine.U.shape == (T, nu) and mine.U.dtype == dtype
