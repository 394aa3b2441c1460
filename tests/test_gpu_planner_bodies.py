"""GPU parity tests (run with ``-m gpu`` on an MI355X), through the C ABI of libnlc_hip.so via the drop-in Python mirror:
the rollout bodies against each other: wave-per-tile / latency-split / one-launch fused, horizon chunks, the staged de Hoog and linear paths, cooperative GRU.  Helpers and tolerances: tests/gpu_common.py.
"""

import glob
import os

import numpy as np
import pytest
import torch

from gpu_common import *  # noqa: F401,F403
from gpu_common import GOLD, TOL, T64, load_sd, build_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env,K,T,h", [("oderl-cartpole", 1024, 20, 128), ("oderl-acrobot", 4096, 12, 128),
                                        ("oderl-pendulum", 16400, 6, 128), ("oderl-cartpole", 2048, 40, 64),
                                        ("oderl-pendulum", 1000, 40, 256), ("oderl-acrobot", 600, 9, 64)])
@pytest.mark.fp64_bit_identity
def test_rollout_kernel_variants_agree(nlc, env, K, T, h):
    """Wave-per-tile (1), latency-split (2: 4 waves per 16-sample tile, LDS exchange) and fused one-launch (3: GRU encode
    and split rollout as roles of one persistent grid, latents handed over inside the launch) rollout bodies: same
    numbers; 2 and 3 share every arithmetic instruction, so they must agree bit for bit."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(0, d, nu, h, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    torch.manual_seed(4)
    raw = torch.randn(K, T, nu, dtype=torch.float64)
    U0 = torch.randn(T, nu, dtype=torch.float64) * 0.2
    state, ab = nlc.initial_state(env), torch.randn(4, nu, dtype=torch.float64)
    out = {}
    for variant in ("1", "2", "3"):
        mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu",
                             lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone(),
                             planner_options={"rollout_variant": int(variant)})
        mppi.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
        act = mppi.command(state, ab)
        out[variant] = (act.clone(), mppi.states.clone(), mppi.cost_total.clone())
    for a, b in zip(out["1"], out["2"]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-11, atol=1e-11)
    for a, b in zip(out["2"], out["3"]):
        assert torch.equal(a, b)


def test_horizon_chunks_pipeline_bit_identical(nlc):
    """Round 3: GRU encode of later horizon chunks on a stream of its own beside the rollout of earlier chunks (wave-per-tile
    body, K > 8192).  The rollout carries state and cost sums between its chunk launches exactly, the encoder's windows do
    not depend on the chunking: same bits as the single launch, over consecutive commands (the chunks of one command must
    also not run into the next command's sampling)."""
    from oracle import nl_model as onl

    env, K, T = "oderl-cartpole", 16384 + 48, 40
    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    planners = {C: nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                                 u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=9,
                                 U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options={"horizon_chunks": C})
                for C in (1, 2, 3, 8)}
    state, ab = nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
    for step in range(4):
        acts = {C: p.command(state, ab) for C, p in planners.items()}
        for C in (2, 3, 8):
            assert torch.equal(acts[1], acts[C]), (C, step)
            for attr in ("states", "cost_total", "omega", "U"):
                assert torch.equal(getattr(planners[1], attr), getattr(planners[C], attr)), (C, attr, step)
        ab = torch.roll(ab, -1, 0)
        ab[-1] = acts[1]


@pytest.mark.parametrize("K,cap,sched", [(2048, 0, None), (2048, 40, None), (1000, 0, None), (4096, 0, None), (600, 7, None),
                                         (16, 0, None), (2048, 200, None), (2048, 0, (0, -1)), (2048, 0, (2, 1)),
                                         (2048, 40, (3, 0)), (1000, 0, (1, 2)), (600, 7, (1, 0)), (4096, 0, (0, 1))])
@pytest.mark.fp64_bit_identity
def test_fused_plan_handoff_repeated_commands(nlc, K, cap, sched):
    """The fused body hands every 16-sample tile's GRU latents from an encoder wavefront to a rollout workgroup INSIDE
    the launch (write-through stores + flag, sc1 loads behind a barrier).  A stale or early read would show up as a
    difference to the two-launch path: 25 consecutive commands (the latent buffer is rewritten in place every command,
    so a stale line of the previous command is a wrong value), all states / costs / actions bit-identical.  cap = 40
    starts only 40 rollout workgroups at the census: the other tiles drain after the encoders, beside busy CUs.
    K = 600 / 16: the encoder ticket is dry almost at once, so census and drain workgroups race for the rollout tiles
    (exclusive owner words); cap = 200: more census rollouts than half the CUs.  sched = (fused_chain_first_tiles,
    fused_partner_tiles): None = the library's auto schedule (one tile first / partner sleeps after two at K = 2048)."""
    from oracle import nl_model as onl

    env, T = "oderl-cartpole", 40
    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    planners = {}
    for variant in (2, 3):
        opts = {"rollout_variant": variant}
        if variant == 3 and cap:
            opts["fused_roll_cap"] = cap
        if variant == 3 and sched is not None:
            opts["fused_chain_first_tiles"], opts["fused_partner_tiles"] = sched
        planners[variant] = nlc.MPPIDelay(
            nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cuda", lambda_=1.0,
            u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=3, planner_options=opts,
        )
    planners[3].U = planners[2].U
    state, ab = nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
    for step in range(25):
        acts = {v: p.command(state, ab) for v, p in planners.items()}
        assert torch.equal(acts[2], acts[3]), step
        assert torch.equal(planners[2].states, planners[3].states), step
        assert torch.equal(planners[2].cost_total, planners[3].cost_total), step
        ab = torch.roll(ab, -1, 0)
        ab[-1] = acts[2].cpu()


@pytest.mark.parametrize("K,env,sample_null", [(2048, "oderl-cartpole", False), (1000, "oderl-acrobot", True),
                                               (4096, "oderl-pendulum", False), (48, "oderl-cartpole", True)])
@pytest.mark.fp64_bit_identity
def test_fused_inline_sampling_and_weights_bit_identical(nlc, K, env, sample_null):
    """Round 3: with device noise the fused body also samples / bounds the actions (encoder role) and reduces the importance
    weights (after the last rollout tile) INSIDE its launch -- command() = that launch + merge_kernel.  Everything the
    command produces must equal, bit for bit, what the launch-per-step bodies produce: the two-launch body (2) and the fused
    body behind its own perturb / weight launches (fused_inline = 0).  12 consecutive commands: perturbed actions, bounded
    noise, actions, states, costs, weights, omega, U and the returned action; a 5-row action buffer and nu = 2 included."""
    from oracle import nl_model as onl

    T = 40 if K > 100 else 9
    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    B = 5 if env == "oderl-pendulum" else 4
    planners = {}
    for name, opts in (("two", {"rollout_variant": 2}), ("fused_sep", {"rollout_variant": 3, "fused_inline": 0}),
                       ("fused_inl", {"rollout_variant": 3, "fused_inline": 1}),
                       ("fused_w3", {"rollout_variant": 3, "fused_blocks_per_cu": 3}),
                       ("fused_w4", {"rollout_variant": 3, "fused_blocks_per_cu": 4, "fused_inline": 2})):
        planners[name] = nlc.MPPIDelay(
            nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=0.7,
            u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=11,
            sample_null_action=sample_null, planner_options=opts, U_init=torch.zeros(T, nu, dtype=torch.float64),
        )
    state = nlc.initial_state(env)
    ab = torch.randn(B, nu, dtype=torch.float64)
    for step in range(12):
        acts = {n: p.command(state, ab) for n, p in planners.items()}
        ref = planners["two"]
        for n in ("fused_sep", "fused_inl", "fused_w3", "fused_w4"):
            p = planners[n]
            assert torch.equal(acts["two"], acts[n]), (n, step)
            for attr in ("perturbed_action", "noise", "actions", "states", "cost_total", "cost_total_non_zero", "omega", "U"):
                assert torch.equal(getattr(ref, attr), getattr(p, attr)), (n, attr, step)
        ab = torch.roll(ab, -1, 0)
        ab[-1] = acts["two"]
    kernels = {n: set(p.ctx.profile_read()) for n, p in planners.items()}  # (empty: profiling is off) -- names checked below
    for n, p in planners.items():
        p.ctx.profile_reset()
        p.ctx.profile(True)
        p.command(state, ab)
        p.ctx.profile(False)
        kernels[n] = set(p.ctx.profile_read())
    assert kernels["fused_inl"] == {"nl_plan_fused_kernel", "merge_kernel"}, kernels
    assert kernels["fused_sep"] == {"perturb_kernel", "nl_plan_fused_kernel", "weight_kernels", "merge_kernel"}, kernels


def test_fused_timeout_reruns_command_on_two_launch_body(nlc):
    """a hand-off time-out of the fused body must not lose the command.  `fused_test_drop_tile` keeps one
    encoder tile from ever being published, so a rollout workgroup gives up after `fused_spin_limit` polls; nlc_mppi_finish
    then re-runs the command on the two-launch body (same inputs, the control sequence before the shift) -- the action
    equals the one a two-launch planner returns, and the ctx stays on the two-launch body afterwards."""
    from oracle import nl_model as onl

    env, K, T = "oderl-cartpole", 512, 12
    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)

    def make(opts):
        return nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                             u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=5,
                             planner_options=opts, U_init=torch.zeros(T, nu, dtype=torch.float64))

    ref = make({"rollout_variant": 2})
    bad = make({"rollout_variant": 3, "fused_test_drop_tile": 37, "fused_spin_limit": 3000})
    state, ab = nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
    for step in range(3):
        a_ref, a_bad = ref.command(state, ab), bad.command(state, ab)
        assert torch.equal(a_ref, a_bad), step
        assert torch.equal(ref.cost_total, bad.cost_total) and torch.equal(ref.U, bad.U), step
        ab = torch.roll(ab, -1, 0)
        ab[-1] = a_ref
    bad.ctx.profile_reset()
    bad.ctx.profile(True)
    bad.command(state, ab)
    bad.ctx.profile(False)
    assert "nl_plan_fused_kernel" not in bad.ctx.profile_read(), "the ctx must stay on the two-launch body after a time-out"
    # the give-up is not silent (ABI v9, nlc_get_stat): one lost launch, one re-run, in command 0; the body in use now
    assert (bad.fused_timeouts, bad.fused_fallbacks, bad.rollout_body) == (1, 1, "latency-split")
    assert int(bad.ctx.get_stat("last_giveup_command")) == 0 and int(bad.ctx.get_stat("fused_lost")) == 1
    assert (ref.fused_timeouts, ref.fused_fallbacks, ref.rollout_body) == (0, 0, "latency-split")
    with pytest.raises(nlc._lib.NlcError):
        bad.ctx.get_stat("no_such_counter")


@pytest.mark.parametrize("algo,S,K,h", [("fixed_tablot", 17, 2500, 128), ("stehfest", 12, 700, 128), ("fixed_tablot", 9, 16500, 128),
                                        ("fixed_tablot", 11, 900, 64), ("stehfest", 8, 16400, 64), ("fixed_tablot", 13, 300, 256)])
def test_linear_ilt_models_on_rollout_kernels_vs_staged_path(nlc, algo, S, K, h):
    """fixed_tablot / stehfest models (hidden width 64 / 128 / 256) plan on the LIN instances of the rollout kernels (the reconstruction
    as two MFMAs per slot group in the epilogue; K <= 8192 the latency-split kernel, above it the wave-per-tile one) -- against
    the staged path (option linear_fused = 0: representation kernel -> slot-major linear ILT -> tail per step), whose sum runs
    in another order, and over two commands."""
    from oracle import nl_model as onl

    env, A = "oderl-cartpole", 3.0
    st = onl.ENV_STATS[env]
    d, nu = st["d"], st["nu"]
    sd = onl.make_synthetic_state_dict(7, d, nu, h, S, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd, S=S, algo=algo)
    T = 9
    g = torch.Generator().manual_seed(4)
    raws = [torch.randn(K, T, nu, dtype=torch.float64, generator=g) for _ in range(2)]
    U0 = torch.randn(T, nu, dtype=torch.float64, generator=g) * 0.2
    state, ab = nlc.initial_state(env), torch.randn(4, nu, dtype=torch.float64, generator=g)
    outs = {}
    for key, opts in (("kernels", {}), ("staged", {"linear_fused": 0})):
        m = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                          u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone(), planner_options=opts)
        assert m.fused
        m.noise_dist = _Replay(*[r.clone() for r in raws])
        with torch.no_grad():
            acts = [m.command(state, ab).clone() for _ in range(2)]
        m.ctx.profile(True)
        with torch.no_grad():
            m.noise_dist = _Replay(raws[0].clone())
            m.command(state, ab)
        names = set(m.ctx.profile_read())
        m.ctx.profile(False)
        assert ("ilt_linear_slot_kernel" in names) == (key == "staged") and ("nl_rollout_kernel" in names) == (key == "kernels")
        outs[key] = (acts, m.states.clone(), m.cost_total.clone(), m.U.clone())
    # (both algorithms sum terms with large alternating weights, and these random weights let some rollouts run away: the two
    # summation orders are compared on the scale of the largest entry, at the north-star bar.
    # U and the action are a softmax over ABSOLUTE cost differences -- of run-away costs of 1e10 here -- and say nothing.)
    for a_, b_ in zip(outs["kernels"][1:3], outs["staged"][1:3]):  # rollout states, total costs (second command)
        sc = float(b_.abs().max()) + 1e-300
        np.testing.assert_allclose(a_.numpy() / sc, b_.numpy() / sc, rtol=0, atol=1e-5)


def test_cfg5_dehoog_planner_staged_hip_path(nlc):
    """BASELINE configs[4] ablation: a de Hoog (33 terms) model plans through the staged all-HIP path
    (rep-func kernel -> de Hoog kernel -> state/cost kernel per horizon step)."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    env, K, T, A, d, nu = "oderl-cartpole", 128, 6, 3.0, 5, 1
    st = onl.ENV_STATS[env]
    sd = onl.make_synthetic_state_dict(2, d, nu, 128, 33, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd, S=33, algo="dehoog")
    sig = nlc.noise_sigma(nu)
    torch.manual_seed(3)
    raw = torch.randn(K, T, nu, dtype=torch.float64)
    U0 = torch.randn(T, nu, dtype=torch.float64) * 0.2
    state, ab = nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, sig, K, T, "cpu", lambda_=1.0,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone())
    assert mppi.fused
    mppi.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
    with torch.no_grad():
        act = mppi.command(state, ab)
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    ref = omppi.mppi_command(U0.clone(), state, ab, raw.clone(), onl.nl_dynamics(sd, ts, S=33, ilt_algorithm="dehoog"),
                             oenvs.RUNNING_COST[env], d, torch.inverse(sig), 1.0, A, torch.tensor(-A), torch.tensor(A))
    np.testing.assert_allclose(mppi.states.numpy(), ref["states"].numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(act.numpy(), ref["action"].numpy(), rtol=1e-5, atol=1e-6)


def test_dehoog_planner_parts_on_streams_bit_identical(nlc):
    """Staged de Hoog planner, round 3: the population cut into P contiguous parts whose launches run on P streams
    (`dehoog_streams`).  A sample's chain never leaves its part and no kernel's per-sample arithmetic depends on the launch
    shape, so P = 1 / 2 / 3 / 4 must give the same bits -- states, costs, weights, action -- over consecutive commands; ragged
    K (the last part is shorter, a part boundary inside a 64-sample QD block is impossible by construction)."""
    from oracle import nl_model as onl

    env, K, T, S = "oderl-cartpole", 4416 + 37, 9, 33
    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(6, d, nu, 128, S, st["state_std"], [A / 2], tame="dehoog")
    model = build_model(nlc, sd, S=S, algo="dehoog")
    planners = {P: nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                                 u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=4,
                                 U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options={"dehoog_streams": P, "dehoog_chain": 0})
                for P in (1, 2, 3, 4)}
    # + the GRU encode in horizon chunks on a stream of its own, beside the chains (cooperative kernel: same bits)
    planners[5] = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                                u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=4,
                                U_init=torch.zeros(T, nu, dtype=torch.float64),
                                planner_options={"dehoog_streams": 2, "dehoog_gru_chunks": 4, "dehoog_chain": 0})
    state, ab = nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64)
    for step in range(3):
        acts = {P: p.command(state, ab) for P, p in planners.items()}
        for P in (2, 3, 4, 5):
            assert torch.equal(acts[1], acts[P]), (P, step)
            for attr in ("states", "cost_total", "omega", "U", "perturbed_action"):
                assert torch.equal(getattr(planners[1], attr), getattr(planners[P], attr)), (P, attr, step)
        ab = torch.roll(ab, -1, 0)
        ab[-1] = acts[1]


@pytest.mark.parametrize("env,K,T,S,per_sample,ext_cost", [
    ("oderl-cartpole", 16384, 40, 33, False, False),   # BASELINE configs[4] at its own size: 256 blocks of 64 samples
    ("oderl-cartpole", 1000 + 37, 7, 33, False, False),  # ragged: the last block holds 13 samples
    ("oderl-acrobot", 70, 5, 33, True, False),         # d = 6 (25 layer-3 tiles, six QD wavefronts), nu = 2, per-sample start states
    ("oderl-pendulum", 200, 6, 17, False, True),       # d = 3, 17 terms, the running cost a caller's closure
    ("oderl-cartpole", 5, 3, 17, False, False),        # fewer samples than one tile
])
@pytest.mark.parametrize("form", [1, 2])
def test_dehoog_step_chain_kernel_bit_identical_to_staged_path(nlc, env, K, T, S, per_sample, ext_cost, form):
    """Round 4: the de Hoog planner's step chain as ONE persistent launch (kernels_dehoog_chain.hip: a workgroup owns 64
    samples for all T steps; representation MLP, QD table and state / cost tail are the staged path's own device functions)
    against the staged 2 T + 1 launches (`dehoog_chain` 1 / 0): states, costs, weights, U and actions must be the same BITS over
    consecutive commands, and the chain planner must not launch the per-step kernels at all.  `form` 1: eight waves own 64
    samples; 2: four waves own 32 samples, two workgroups per CU, two dims' QD rows per wavefront."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(6, d, nu, 128, S, st["state_std"], [A / 2], tame="dehoog")
    model = build_model(nlc, sd, S=S, algo="dehoog")
    if ext_cost:
        def cost(x, u):
            return (x * x).sum(-1) * 0.3 + 0.01 * (u * u).sum(-1)
    else:
        cost = nlc.EnvCost(env)
    planners = {ch: nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), cost, d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                                  u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=9,
                                  U_init=torch.zeros(T, nu, dtype=torch.float64),
                                  planner_options={"dehoog_chain": form if ch else 0})
                for ch in (0, 1)}
    g = torch.Generator().manual_seed(3)
    state = nlc.initial_state(env, g)
    if per_sample:
        state = state + 0.01 * torch.randn(K, d, dtype=torch.float64, generator=g)
    ab = torch.zeros(4, nu, dtype=torch.float64)
    planners[1].ctx.profile(True)
    for step in range(2 if K > 10000 else 3):
        with torch.no_grad():
            acts = {ch: p.command(state, ab) for ch, p in planners.items()}
        for attr in ("perturbed_action", "states", "cost_total", "omega", "U"):
            x0, x1 = getattr(planners[0], attr), getattr(planners[1], attr)
            assert torch.equal(x0, x1), (attr, step, float((x0 - x1).abs().max()))
        assert torch.equal(acts[0], acts[1]) and bool(torch.isfinite(acts[1]).all()), step
        ab = torch.roll(ab, -1, 0)
        ab[-1] = acts[0].view(-1, nu)[0]
    planners[1].ctx.profile(False)
    prof = planners[1].ctx.profile_read()
    assert "nl_dehoog_chain_kernel" in prof and "ilt_dehoog_kernel" not in prof and "nl_repfunc_kernel" not in prof, sorted(prof)


def test_repfunc_split_kernel_agrees_with_wave_per_tile_planner(nlc):
    """Staged de Hoog planner: the latency-split representation launch (one workgroup per 16-sample tile) against the
    wave-per-tile one -- same GEMM order per output tile and the same sphere map, so the two agree to rounding of the
    differently contracted scalar code (1e-9 after the QD recurrence's amplification; the oracle comparisons of the
    de Hoog tests run through the split form, the default) -- ragged K, several term counts."""
    from oracle import nl_model as onl

    for env, S, K, T in (("oderl-cartpole", 33, 1000, 6), ("oderl-acrobot", 9, 333, 4), ("oderl-pendulum", 21, 16, 3)):
        st = onl.ENV_STATS[env]
        d, nu, A = st["d"], st["nu"], st["act_high"]
        sd = onl.make_synthetic_state_dict(6, d, nu, 128, S, st["state_std"], [A / 2], tame="dehoog")
        model = build_model(nlc, sd, S=S, algo="dehoog")
        state0 = nlc.initial_state(env, torch.Generator().manual_seed(2))
        res = []
        for split in (0, 1):
            p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cuda", lambda_=1.0,
                              u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=4,
                              U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options={"repfunc_split": split})
            with torch.no_grad():
                acts = [p.command(state0, torch.zeros(4, nu, dtype=torch.float64)).cpu() for _ in range(2)]
            res.append((torch.stack(acts), p.states.cpu(), p.cost_total.cpu()))
        for a, b in zip(res[0], res[1]):
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-9, atol=1e-10, err_msg=f"{env} S={S}")


@pytest.mark.fp64_bit_identity
def test_gru_cooperative_kernel_bit_identical(nlc):
    """gru_encode_coop_kernel (one 16-window tile per workgroup, one gate chunk per wavefront; what small launches and the
    fused body's encoders run) against the wave-per-tile kernel: same chunk GEMMs in the same k order and the same gate
    math, so every latent is bit-identical -- ragged N, both kernels forced through the option, and a two-launch planner
    command with either."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS["oderl-acrobot"]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    torch.manual_seed(9)
    for h in (128, 64, 256):  # GRU width 64 (one chunk per wave), 32 (two waves idle), 128 (two chunks per wave)
        sd = onl.make_synthetic_state_dict(3, d, nu, h, 17, st["state_std"], [A / 2], tame=True)
        model = build_model(nlc, sd)
        ctx = model.hip_ctx(torch.device("cuda:0"))
        try:
            for N, B in ((1, 4), (15, 4), (16, 5), (17, 4), (1000, 3), (40961, 4)):
                win = ((torch.rand(N, B, nu, dtype=torch.float64) * 2 - 1) * A).cuda()
                outs = []
                for coop in (0, 1):
                    ctx.set_option("gru_coop", coop)
                    with torch.no_grad():
                        outs.append(model.encode_actions(win).clone())
                assert torch.equal(outs[0], outs[1]), (h, N, B)
        finally:
            ctx.set_option("gru_coop", -1)
    acts = []
    state0 = nlc.initial_state("oderl-acrobot", torch.Generator().manual_seed(2))
    for coop in (0, 1):
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-acrobot"), d, nlc.noise_sigma(nu), 700, 9, "cuda",
                          lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=4,
                          U_init=torch.zeros(9, nu, dtype=torch.float64), planner_options={"rollout_variant": 2, "gru_coop": coop})
        with torch.no_grad():
            acts.append((p.command(state0, torch.zeros(4, nu, dtype=torch.float64)).cpu(), p.states.cpu()))
    assert torch.equal(acts[0][0], acts[1][0]) and torch.equal(acts[0][1], acts[1][1])
