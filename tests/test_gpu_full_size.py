"""GPU parity tests (run with ``-m gpu`` on an MI355X), through the C ABI of libnlc_hip.so via the drop-in Python mirror:
every BASELINE config at its whole population (oracle on a strided subset, seed replay of the reference fixtures), random and edge shapes.  Helpers and tolerances: tests/gpu_common.py.
"""

import glob
import os

import numpy as np
import pytest
import torch

from gpu_common import *  # noqa: F401,F403
from gpu_common import GOLD, TOL, T64, load_sd, build_model

pytestmark = pytest.mark.gpu


def test_full_size_cfg2_properties(nlc, encoder_mode):
    """K=16384, T=40 cartpole, NL dynamics: a sample subset against the oracle + size-independent properties."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    env, K, T, A, d, nu = "oderl-cartpole", 16384, 40, 3.0, 5, 1
    st = onl.ENV_STATS[env]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    sig = nlc.noise_sigma(nu)
    torch.manual_seed(0)
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, sig, K, T, "cuda", lambda_=1.0,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)
    state, ab = nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
    U_before = mppi.U.cpu()
    action = mppi.command(state, ab.cuda())
    V, eps = mppi.perturbed_action.cpu(), mppi.noise.cpu()
    # (1) sample subset through the oracle (same bounded actions): states after 40 sequential steps
    idx = torch.arange(0, K, K // 128)
    ts = torch.full((len(idx), 1), 0.05, dtype=torch.float64)
    cost_ref, states_ref, _ = omppi.rollout(state, ab, V[idx], A, onl.nl_dynamics(sd, ts, S=17),
                                            oenvs.RUNNING_COST[env], d)
    np.testing.assert_allclose(mppi.states.cpu()[idx].numpy(), states_ref.numpy(), rtol=1e-7, atol=1e-7)
    U_shift = torch.roll(U_before, -1, 0)
    U_shift[-1] = 0
    pc = torch.sum(U_shift * (eps[idx] @ torch.inverse(sig)), dim=(1, 2))
    np.testing.assert_allclose(mppi.cost_total.cpu()[idx].numpy(), (cost_ref + pc).numpy(), rtol=1e-7, atol=1e-7)
    # (2) properties over the whole population
    assert torch.all(V.abs() <= 1.0 + 1e-15)  # bounded to [-A, A]/A
    np.testing.assert_allclose((U_shift + eps).clamp(-1, 1).numpy(), V.numpy(), rtol=0, atol=1e-15)
    omega = mppi.omega.cpu()
    assert abs(float(omega.sum()) - 1.0) < 1e-12 and float(mppi.cost_total_non_zero.max()) == 1.0
    cost = mppi.cost_total.cpu()
    w = torch.exp(-(cost - cost.min()))
    np.testing.assert_allclose(omega.numpy(), (w / w.sum()).numpy(), rtol=1e-10, atol=1e-16)
    U_after = U_shift + torch.einsum("k,ktj->tj", omega, eps)
    np.testing.assert_allclose(mppi.U.cpu().numpy(), U_after.numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(action.cpu().numpy(), (U_after[0] * A).numpy(), rtol=1e-10, atol=1e-13)


def test_cfg1_cartpole_1024x20(nlc, encoder_mode):
    """BASELINE configs[0] shape: K=1024, H=20 (latency-split rollout kernel)."""
    _subset_check(nlc, "oderl-cartpole", 1024, 20, 4, n_check=128)


def test_cfg3_pendulum_shard_32768x40_window5(nlc, encoder_mode):
    """BASELINE configs[2] per-GPU shard: pendulum, 65536/2 samples, H=40, action_buffer_size=5 (delay 4)."""
    _subset_check(nlc, "oderl-pendulum", 32768, 40, 5)


def test_cfg4_acrobot_shard_32768x60(nlc, encoder_mode):
    """BASELINE configs[3] per-GPU shard: acrobot (nx=6, nu=2), 262144/8 samples, H=60."""
    _subset_check(nlc, "oderl-acrobot", 32768, 60, 4)


def test_cfg4_acrobot_whole_population_262144x60(nlc, encoder_mode):
    """BASELINE configs[3] at its WHOLE size on one GPU (the largest population any config names): acrobot, K = 262144,
    H = 60, nu = 2 -- 64 strided samples through the oracle, weights / U / action over all 262144."""
    _subset_check(nlc, "oderl-acrobot", 262144, 60, 4)
    torch.cuda.empty_cache()


def test_cfg3_pendulum_whole_population_65536x40(nlc, encoder_mode):
    """BASELINE configs[2] at its whole size on one GPU: pendulum, K = 65536, H = 40, 5-row action buffer."""
    _subset_check(nlc, "oderl-pendulum", 65536, 40, 5)
    torch.cuda.empty_cache()


def test_full_size_cfg5_dehoog(nlc, encoder_mode):
    """BASELINE configs[4] at its own size: cartpole, de Hoog ILT with 33 terms, K = 16384, T = 40, on the staged
    all-HIP path.  64 strided samples through the oracle (mpmath's de Hoog recurrences with IEEE divisions; the kernel
    divides by a refined reciprocal inside the QD table): states after 40 sequential steps and costs must meet the
    north-star bar of 1e-5; softmax weights, U and the action are checked over the whole population.
    Weights: the "trained-like" de Hoog model of oracle.nl_model.tame_dehoog_ (F(s_k) a perturbed Laplace transform;
    with the Fourier models' phi-shifted random weights the QD table hits near-poles and a 1e-10 perturbation of the
    state grows to O(100) by T = 40 in the ORACLE itself -- nothing to compare)."""
    _subset_check(nlc, "oderl-cartpole", 16384, 40, 4, S=33, algo="dehoog", tol=1e-5, tame="dehoog")


@pytest.mark.parametrize("algo,S", [("fixed_tablot", 17), ("stehfest", 16)])
def test_full_size_linear_ilt_models(nlc, algo, S):
    """The other closed-form values of nl_ilt_algorithm at configs[1]'s size (cartpole, K = 16384, T = 40) on the staged
    all-HIP path (representation kernel -> slot-major linear reconstruction -> state / cost tail per horizon step): 64
    strided samples through the oracle after 40 sequential steps at the north-star bar, softmax weights / U / action over
    the whole population.  (Both algorithms sum terms with large alternating weights, so last-bit differences of F_k come
    back amplified: the sweep's short-horizon cases hold 2e-5 / 1e-6.)"""
    _subset_check(nlc, "oderl-cartpole", 16384, 40, 4, S=S, algo=algo, tol=1e-5)


def _random_shape_cases(n=100, seed=2024):
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(n):
        env = ["oderl-cartpole", "oderl-pendulum", "oderl-acrobot"][rng.randint(3)]
        h = [64, 128, 256][rng.randint(3)]
        algo = ["fourier", "fourier", "dehoog", "fixed_tablot", "stehfest"][rng.randint(5)]
        if algo == "fourier":
            S = int(rng.randint(3, 34))
        elif algo == "stehfest":
            S = int(2 * rng.randint(2, 8))
        else:
            S = int(2 * rng.randint(1, 17) + 1)
        cases.append((i, env, h, algo, S, int(rng.randint(1, 7)), int(rng.randint(1, 11)), int(rng.randint(1, 400))))
    # the smallest problems there are: one sample, one step, a one-row window; two samples over the full horizon
    cases.append((n, "oderl-cartpole", 128, "fourier", 17, 1, 1, 1))
    cases.append((n + 1, "oderl-acrobot", 128, "fourier", 17, 4, 40, 2))
    cases.append((n + 2, "oderl-pendulum", 128, "dehoog", 33, 2, 1, 1))
    # cases of the kind the one-off 123-case run of round 3 lost (width-256 fixed Talbot / Stehfest models whose rollouts run away
    # over 9-10 steps): ill-conditioned in the ORACLE itself -- the condition-aware bound below has to carry them
    cases.append((n + 3, "oderl-acrobot", 256, "fixed_tablot", 31, 4, 10, 300))
    cases.append((n + 4, "oderl-cartpole", 256, "stehfest", 14, 3, 10, 350))
    cases.append((n + 5, "oderl-pendulum", 256, "fixed_tablot", 33, 5, 9, 250))
    return cases


_PROBE = 1e-13  # relative size of the input perturbation that measures the oracle's own sensitivity
_PROBE_FACTOR = 200.0  # a float64 implementation may differ from another by this many probe responses


def _sweep_case(nlc, i, env, h, algo, S, B, T, K, planner_options=None):
    """One planning step on the GPU and through the oracle -- twice: the second oracle run has its inputs (start state, raw
    noise draw) perturbed by a relative 1e-13, which measures how far THIS problem amplifies last-bit differences (fixed Talbot /
    Stehfest weights alternate at ~e^{0.4 S}; untamed rollouts can run away).  Returns (planner, reference, per-sample per-step
    sensitivity of the states, sensitivity of the costs, sensitivity of the action)."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(100 + i, d, nu, h, S, st["state_std"], [A / 2], tame="dehoog" if algo == "dehoog" else True)
    model = build_model(nlc, sd, S=S, algo=algo)
    torch.manual_seed(500 + i)
    sig = nlc.noise_sigma(nu)
    raw = torch.randn(K, T, nu, dtype=torch.float64) @ torch.linalg.cholesky(sig).T
    U0 = torch.randn(T, nu, dtype=torch.float64) * 0.2
    state, ab = nlc.initial_state(env), torch.randn(B, nu, dtype=torch.float64) * 0.3
    tsk = torch.full((K, 1), 0.05, dtype=torch.float64)

    def oracle(state_, raw_):
        return omppi.mppi_command(U0.clone(), state_, ab, raw_.clone(), onl.nl_dynamics(sd, tsk, S=S, ilt_algorithm=algo),
                                  oenvs.RUNNING_COST[env], d, torch.inverse(sig), 1.0, A, torch.tensor(-A), torch.tensor(A))

    ref = oracle(state, raw)
    gen = torch.Generator().manual_seed(900 + i)
    sgn = lambda t: torch.where(torch.rand(t.shape, generator=gen) < 0.5, -1.0, 1.0).to(torch.float64)  # noqa: E731
    ref_p = oracle(state * (1.0 + _PROBE * sgn(state)), raw * (1.0 + _PROBE * sgn(raw)))
    # error grows along the horizon: a sample's bound at step t is its largest response up to t
    ds = (ref_p["states"] - ref["states"]).abs().amax(dim=2)                 # (K, T)
    sens_states = torch.cummax(ds, dim=1).values.unsqueeze(-1)               # (K, T, 1)
    sens_cost = (ref_p["cost_total"] - ref["cost_total"]).abs()              # (K)
    sens_act = (ref_p["action"] - ref["action"]).abs().max()
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, sig, K, T, "cpu", lambda_=1.0,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone(),
                         planner_options=planner_options)
    mppi.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
    with torch.no_grad():
        act = mppi.command(state, ab)
    return mppi, act, ref, sens_states, sens_cost, sens_act


def _assert_within_condition(got, want, sens, what, rtol=1e-9, atol=1e-10):
    """|got - want| <= atol + rtol |want| + _PROBE_FACTOR * (the oracle's response to a 1e-13 input perturbation): tight (1e-9)
    where the problem is well conditioned, as wide as the problem's own amplification where it is not."""
    got, want = torch.as_tensor(got, dtype=torch.float64), torch.as_tensor(want, dtype=torch.float64)
    bound = atol + rtol * want.abs() + _PROBE_FACTOR * sens
    err = (got - want).abs()
    bad = err > bound
    if bool(bad.any()) or not bool(torch.isfinite(got).all()):
        worst = int(torch.argmax(err - bound))
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} entries beyond the condition-aware bound; worst |err| "
                             f"{float(err.reshape(-1)[worst]):.3e} vs bound {float(bound.expand_as(err).reshape(-1)[worst]):.3e}")


@pytest.mark.parametrize("i,env,h,algo,S,B,T,K", _random_shape_cases())
def test_random_shape_sweep_planner_vs_oracle(nlc, i, env, h, algo, S, B, T, K):
    """Seeded random shapes (env, hidden width, ILT algorithm and term count, window length B, horizon T, population K --
    ragged against every tile size): one planning step on the auto-selected rollout body against the oracle, 106 cases.
    The tolerance is CONDITION-AWARE: 1e-9 plus a multiple of the oracle's own response to a 1e-13
    relative perturbation of its inputs, so a well-conditioned case is held to 1e-9 (the fixed 2e-5 of round 3 would have
    hidden a real error in a LIN coefficient: test_sweep_bound_catches_a_corrupted_lin_coefficient) while the ill-conditioned
    ones (runaway fixed Talbot / Stehfest rollouts) pass for the stated reason instead of being excluded."""
    mppi, act, ref, sens_states, sens_cost, sens_act = _sweep_case(nlc, i, env, h, algo, S, B, T, K)
    _assert_within_condition(mppi.states, ref["states"], sens_states, "states")
    _assert_within_condition(mppi.cost_total, ref["cost_total"], sens_cost, "cost_total")
    _assert_within_condition(act, ref["action"], sens_act, "action")


@pytest.mark.parametrize("algo,S", [("fixed_tablot", 17), ("stehfest", 12)])
def test_sweep_bound_catches_a_corrupted_lin_coefficient(nlc, algo, S):
    """The condition-aware bound is not a licence: one coefficient of the LIN rollout instances' folded (w_re / t) fragment
    scaled by (1 + 1e-6) at configure time (`test_lin_coeff_scale`, tests only) must FAIL the sweep's comparison, while the
    uncorrupted planner passes it on the same case."""
    case = (7, "oderl-cartpole", 128, algo, S, 4, 8, 200)
    mppi, act, ref, sens_states, sens_cost, sens_act = _sweep_case(nlc, *case)
    _assert_within_condition(mppi.states, ref["states"], sens_states, "states")
    bad, act_b, ref, sens_states, sens_cost, sens_act = _sweep_case(nlc, *case, planner_options={"test_lin_coeff_scale": 1.0 + 1e-6})
    with pytest.raises(AssertionError, match="beyond the condition-aware bound"):
        _assert_within_condition(bad.states, ref["states"], sens_states, "states")


def test_state_dim_4_planner(nlc):
    """SURVEY 8d's literal "state_dim = 4" variant: a 4-dim observation (no trig embedding), nu = 1, NL dynamics in the
    fused rollout, the running cost a caller's closure (no reference env has d = 4): K = 2048, T = 40."""
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    d, nu, K, T, A, S = 4, 1, 2048, 40, 3.0, 17
    sd = onl.make_synthetic_state_dict(7, d, nu, 128, S, [1.0, 2.0, 0.5, 3.0], [A / 2], tame=True)
    model = build_model(nlc, sd, S=S)

    def cost(x, u):
        return (x[..., 0] ** 2 + 0.1 * x[..., 1] ** 2 + (x[..., 2] - 1.0) ** 2 + 0.01 * x[..., 3] ** 2) + 0.01 * (u * u).sum(-1)

    torch.manual_seed(11)
    raw = torch.randn(K, T, nu, dtype=torch.float64)
    U0 = torch.randn(T, nu, dtype=torch.float64) * 0.2
    state, ab = torch.tensor([0.1, -0.2, 0.3, 0.05], dtype=torch.float64), torch.zeros(4, nu, dtype=torch.float64)
    sig = nlc.noise_sigma(nu)
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), cost, d, sig, K, T, "cpu", lambda_=1.0, u_min=torch.tensor(-A),
                         u_max=torch.tensor(A), u_scale=A, U_init=U0.clone())
    assert mppi.fused_dynamics and mppi.cost_external
    mppi.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
    with torch.no_grad():
        act = mppi.command(state, ab)
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    ref = omppi.mppi_command(U0.clone(), state, ab, raw.clone(), onl.nl_dynamics(sd, ts, S=S), cost, d, torch.inverse(sig),
                             1.0, A, torch.tensor(-A), torch.tensor(A))
    np.testing.assert_allclose(mppi.states.numpy(), ref["states"].numpy(), rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(mppi.cost_total.numpy(), ref["cost_total"].numpy(), rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(act.numpy(), ref["action"].numpy(), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("env,K,T,B", [("oderl-cartpole", 5, 1, 1), ("oderl-acrobot", 17, 3, 2), ("oderl-pendulum", 1, 4, 6),
                                       ("oderl-acrobot", 33, 2, 20)])  # B*nu = 40: beyond the kernel-argument staging
def test_tiny_and_ragged_planner_shapes_vs_oracle(nlc, env, K, T, B):
    """K below one MFMA tile, horizon 1, a one-row action buffer (no history) and a long one (B = 6 > default)."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(12, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    g = torch.Generator().manual_seed(K * 100 + T)
    raw = torch.randn(K, T, nu, dtype=torch.float64, generator=g)
    U0 = torch.randn(T, nu, dtype=torch.float64, generator=g) * 0.2
    ab = torch.randn(B, nu, dtype=torch.float64, generator=g)
    state = _state(nlc, env, 3)
    sig = nlc.noise_sigma(nu)
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    for name, dyn_gpu, dyn_ref in (
        ("nl", nlc.NLDynamics(model, 0.05), onl.nl_dynamics(sd, ts, S=17)),
        ("oracle", nlc.OracleDynamics(env, 0.05, B - 1), lambda s, w: oenvs.ORACLE_DYNAMICS[env](s, w, ts, B - 1)),
    ):
        p = nlc.MPPIDelay(dyn_gpu, nlc.EnvCost(env), d, sig, K, T, "cpu", lambda_=1.3, u_min=torch.tensor(-A),
                          u_max=torch.tensor(A), u_scale=A, U_init=U0.clone())
        p.noise_dist = _Replay(raw.clone())
        with torch.no_grad():
            # the reference accepts anything torch.tensor() takes for the state (mppi_delay.py:196-197); a float64
            # ndarray keeps its precision (a python list would become float32 there, and here)
            act = p.command(state.numpy().astype(np.float64), ab)
        ref = omppi.mppi_command(U0.clone(), state, ab, raw.clone(), dyn_ref, oenvs.RUNNING_COST[env], d, torch.inverse(sig),
                                 1.3, A, torch.tensor(-A), torch.tensor(A))
        np.testing.assert_allclose(act.numpy(), ref["action"].numpy(), err_msg=name, **TOL)
        np.testing.assert_allclose(p.states.numpy(), ref["states"].numpy(), err_msg=name, **TOL)
        np.testing.assert_allclose(p.omega.numpy(), ref["omega"].numpy(), err_msg=name, **TOL)
        assert p.U.shape == (T, nu) and p.actions.shape == (K, T, nu)


def test_full_size_cfg2_vs_reference_golden_seed_replay(nlc, encoder_mode):
    """G6: BASELINE configs[1] at full size (K=16384, T=40) against the REAL reference MPPIDelay + NeuralLaplaceModel,
    two consecutive commands.  device="cpu" + torch.manual_seed replays the reference's generator stream (ctor U
    draw, one (K, T) draw per command), so the fixture needs no noise tensor."""
    from oracle import nl_model as onl

    g = np.load(f"{GOLD}/g6_full_cfg2.npz")
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["d"]), int(g["nu"]), float(g["A"])
    st = onl.ENV_STATS["oderl-cartpole"]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    torch.manual_seed(int(g["seed"]))
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-cartpole"), d, nlc.noise_sigma(nu), K, T, "cpu",
                         lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)
    assert mppi.fused
    np.testing.assert_array_equal(mppi.U.numpy(), g["U0"])
    sub = g["sub"]
    with torch.no_grad():
        for step in range(2):
            pre = f"s{step}_"
            act = mppi.command(g[pre + "state"], T64(g[pre + "action_buffer"]))
            np.testing.assert_allclose(act.numpy(), g[pre + "action"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(mppi.U.numpy(), g[pre + "U_after"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(mppi.cost_total.numpy(), g[pre + "cost_total"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(mppi.omega.numpy(), g[pre + "omega"], rtol=1e-7, atol=1e-30)
            np.testing.assert_allclose(mppi.states.numpy()[sub], g[pre + "states_sub"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(mppi.noise.numpy()[sub], g[pre + "noise_sub"], rtol=0, atol=1e-11)


@pytest.mark.parametrize("tag,env", [("cfg1", "oderl-cartpole"), ("cfg3", "oderl-pendulum"), ("cfg4", "oderl-acrobot")])
def test_full_size_cfg1_cfg3_cfg4_vs_reference_golden_seed_replay(nlc, tag, env, encoder_mode):
    """G7: BASELINE configs[0] (cartpole, K=1024, T=20), configs[2] (pendulum, K=65536, T=40, 5-row buffer) and configs[3] (acrobot, K=262144, T=60) at
    their FULL population on one GPU against the real reference (seed replay, see G6); cost/omega/states on a strided
    subset plus the population aggregates beta = min cost, eta = sum of weights, sum of costs."""
    from oracle import nl_model as onl

    g = np.load(f"{GOLD}/g7_full_{tag}.npz")
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["d"]), int(g["nu"]), float(g["A"])
    st = onl.ENV_STATS[env]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    torch.manual_seed(int(g["seed"]))
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu",
                         lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)
    np.testing.assert_array_equal(mppi.U.numpy(), g["U0"])
    sub = g["sub"]
    # the fixture's action buffer came from the same global generator, between the ctor and the command
    B = int(g["B"])
    ab = (torch.rand(B, nu, dtype=torch.float64) - 0.5) * A
    np.testing.assert_array_equal(ab.numpy(), g["action_buffer"])
    with torch.no_grad():
        act = mppi.command(g["state"], ab)
    np.testing.assert_allclose(act.numpy(), g["action"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(mppi.U.numpy(), g["U_after"], rtol=1e-8, atol=1e-10)
    cost = mppi._cost_total  # device tensors: only the subset travels
    np.testing.assert_allclose(float(cost.min()), float(g["beta"]), rtol=1e-10)
    np.testing.assert_allclose(float(mppi._cost_nz.sum()), float(g["eta"]), rtol=1e-8)
    np.testing.assert_allclose(float(cost.sum()), float(g["cost_sum"]), rtol=1e-9)
    idx = torch.as_tensor(sub, device=cost.device)
    np.testing.assert_allclose(cost[idx].cpu().numpy(), g["cost_total_sub"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(mppi._omega[idx].cpu().numpy(), g["omega_sub"], rtol=1e-7, atol=1e-30)
    np.testing.assert_allclose(mppi._states[idx].cpu().numpy(), g["states_sub"], rtol=1e-8, atol=1e-8)
    np.testing.assert_array_equal(mppi._noise[idx].cpu().numpy(), g["noise_sub"])


@pytest.mark.parametrize("env", ["oderl-cartpole", "oderl-acrobot"])
def test_untamed_random_weights_short_horizon(nlc, env):
    """Reference-constructor weights WITHOUT the 'trained-like' phi shift: the model is chaotic (outputs grow ~10x per
    step) and some sphere angles saturate, so this only runs a short horizon and compares relative to the state scale --
    it pins the saturation handling of the fused sphere map against the oracle's torch.tan / torch.tanh."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=False)
    model = build_model(nlc, sd)
    K, T = 96, 4
    g = torch.Generator().manual_seed(8)
    raw = torch.randn(K, T, nu, dtype=torch.float64, generator=g)
    U0 = torch.zeros(T, nu, dtype=torch.float64)
    state, ab = _state(nlc, env, 2), torch.zeros(4, nu, dtype=torch.float64)
    sig = nlc.noise_sigma(nu)
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, sig, K, T, "cpu", lambda_=1.0,
                      u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone())
    p.noise_dist = _Replay(raw.clone())
    with torch.no_grad():
        p.command(state, ab)
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    ref = omppi.mppi_command(U0.clone(), state, ab, raw.clone(), onl.nl_dynamics(sd, ts, S=17), oenvs.RUNNING_COST[env], d,
                             torch.inverse(sig), 1.0, A, torch.tensor(-A), torch.tensor(A))
    got, want = p.states, ref["states"]
    assert torch.isfinite(got).all() and torch.isfinite(want).all()
    for t in range(T):
        scale = float(want[:, t].abs().max())
        err = float((got[:, t] - want[:, t]).abs().max()) / scale
        assert err < 1e-9 * 10.0 ** (3 * t), (t, err, scale)  # chaotic amplification: ~1000x per step at most


@pytest.mark.parametrize("key", ["0", "d4"])
def test_bench_other_configs_emit_a_complete_line(key):
    """`bench.py --config k` measures the other BASELINE configs at their own population with the same contract
    fields, its own roofline (dominant kernel + whole step) and a CPU baseline on a budget; `config.commit` is never null and the
    library's source hash travels with the line."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--config", key, "--steps", "10", "--warmup", "2", "--no-ilt",
                          "--cpu-budget", "8", "--preheat-ms", "50"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    want_K = {"0": 1024, "d4": 16384}[key]
    assert rec["n_gpus"] == 1 and rec["steps"] == 10 and rec["dtype"] == "f64" and rec["vs_baseline"] is None and rec["value"] > 0
    assert f"K={want_K}" in rec["config"]["workload"] and rec["config"]["baseline_config"] == key
    assert rec["config"]["commit"] and rec["config"]["library_build"]["csrc_sha"]
    roof = rec["roofline"]
    assert roof["bound"] == "mfma" and 0.0 < roof["frac"] < 1.0 and 0.0 < roof["step"]["frac"] < 1.0
    assert roof["kernel"] == ("nl_plan_fused_kernel" if key == "0" else "gru_encode_kernel")
    cpu = rec["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and (cpu["value"] or cpu["value_extrapolated"]) and "sample" in cpu
    assert rec["config"]["ranks_seen"]["per_rank"][0]["rollout_body"] == ("fused" if key == "0" else "wave-per-tile")
