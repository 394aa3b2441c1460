"""GPU parity tests (run with ``-m gpu`` on an MI355X), through the C ABI of libnlc_hip.so via the drop-in Python mirror:
Delta-t RNN / RNN / NODE baselines (f4) and the env step / device-resident loop (f3).  Helpers and tolerances: tests/gpu_common.py.
"""

import glob
import os

import numpy as np
import pytest
import torch

from gpu_common import *  # noqa: F401,F403
from gpu_common import GOLD, TOL, T64, load_sd, build_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env,published,random_policy", [("oderl-cartpole", -139.69, -14246.30),
                                                         ("oderl-pendulum", -121.05, -616.77)])
def test_closed_loop_episode_return_near_published_oracle_mpc(nlc, env, published, random_policy):
    """Behavioural check: 200 control steps of the reference's evaluation loop (mppi_with_model.py:244-317) with
    oracle dynamics (K=1000, T=40 as in config.py:52-53), planner on the GPU, the env's Euler step on the host.
    The return must be in the neighbourhood of the reference's published oracle+MPC return
    (process_results/plot_util.py:7-11) -- far from the random-policy return (:2-6)."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi

    nx, nu, A = oenvs.OBS_DIM[env], oenvs.ACT_DIM[env], oenvs.ACTION_HIGH[env]
    mppi = nlc.MPPIDelay(nlc.OracleDynamics(env, 0.05, 0), nlc.EnvCost(env), nx, nlc.noise_sigma(nu), 1000, 40, "cpu",
                         lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                         U_init=torch.zeros(40, nu, dtype=torch.float64))
    obs = oenvs.initial_state(env, 0)
    if env == "oderl-pendulum":
        obs = torch.tensor([-1.0, 0.0, 1.0], dtype=torch.float64)  # harness start [pi, 1] (mppi_with_model.py:188-189)
    ab = torch.zeros(4, nu, dtype=torch.float64)
    ts = torch.full((1, 1), 0.05, dtype=torch.float64)
    total = 0.0
    for _ in range(200):
        a = mppi.command(obs, ab)
        ab, applied = omppi.get_action(ab, a, 0)
        obs = oenvs.ORACLE_DYNAMICS[env](obs.view(1, -1), applied.view(1, 1, nu), ts, 0).view(-1)
        total += -float(oenvs.RUNNING_COST[env](obs.view(1, -1), applied.view(1, nu)))
    assert 1.6 * published < total < 0.6 * published, (total, published)
    assert total > 0.5 * random_policy


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_dtrnn_forward_vs_reference_golden(nlc, env):
    """G9: HIP DeltaTRNN.forward vs the REAL reference class (train_utils.py:589-631), both input branches."""
    g = np.load(f"{GOLD}/g9_dtrnn_{env}.npz")
    sd = load_sd(g, "sd_")
    obs, win, ts = T64(g["fwd_obs"]), T64(g["fwd_window"]), T64(g["fwd_ts"])
    with torch.no_grad():
        got = build_rnn(nlc, sd, int(g["H"]))(obs.cuda(), win.cuda(), ts.cuda()).cpu()
        np.testing.assert_allclose(got.numpy(), g["fwd_out"], **TOL)
        raw = build_rnn(nlc, sd, int(g["H"]), normalize_time=False)(obs, win, ts)  # CPU inputs -> CPU result
        assert raw.device.type == "cpu"
        np.testing.assert_allclose(raw.numpy(), g["raw_out"], **TOL)
        with pytest.raises(NameError):
            build_rnn(nlc, sd, int(g["H"]), normalize=False, normalize_time=True)(obs, win, ts)
    frozen = build_rnn(nlc, sd, int(g["H"]))
    for p_ in frozen.parameters():
        p_.requires_grad_(False)
    with pytest.raises(NotImplementedError):
        frozen(obs, win, ts)  # grad mode with nothing to train: the HIP path is inference-only
    # grad mode with trainable parameters: the same op sequence on PyTorch-ROCm, gradients = autograd of the oracle
    from oracle import rnn_model as ornn

    leaves = {k: (v.clone().requires_grad_() if k.startswith(("gru.", "linear_out.")) else v) for k, v in sd.items()}
    ref = ornn.forward(leaves, obs, win, ts)
    ref.square().sum().backward()
    m = build_rnn(nlc, sd, int(g["H"]))
    out = m(obs.cuda(), win.cuda(), ts.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["fwd_out"], **TOL)
    out.square().sum().backward()
    for k, p_ in m.named_parameters():
        sc = float(leaves[k].grad.abs().max()) + 1e-300
        np.testing.assert_allclose(p_.grad.cpu().numpy() / sc, leaves[k].grad.numpy() / sc, rtol=1e-7, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("env", ["cartpole", "acrobot"])
def test_plain_rnn_baseline_vs_reference_golden(nlc, env):
    """G9: the plain RNN baseline (train_utils.py:550-586) vs the REAL reference class on both input branches, and
    behind the planner vs the CPU oracle."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import rnn_model as ornn

    g = np.load(f"{GOLD}/g9_dtrnn_{env}.npz")
    sd = load_sd(g, "rnnsd_")
    d, nu, A = int(g["nx"]), int(g["nu"]), float(g["A"])
    obs, win, ts = T64(g["fwd_obs"]), T64(g["fwd_window"]), T64(g["fwd_ts"])

    def build(normalize):
        m = nlc.RNN(d, nu, hidden_units=64, state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0]),
                    action_std=np.array([1.0]), normalize=normalize).double()
        m.load_state_dict(sd)
        return m.cuda()

    with torch.no_grad():
        np.testing.assert_allclose(build(True)(obs.cuda(), win.cuda(), ts.cuda()).cpu().numpy(), g["rnn_out"], **TOL)
        np.testing.assert_allclose(build(False)(obs, win, ts).numpy(), g["rnn_raw_out"], **TOL)
    K, Tt = 80, 6
    gen = torch.Generator().manual_seed(31)
    raw = torch.randn(K, Tt, nu, dtype=torch.float64, generator=gen)
    U0 = torch.randn(Tt, nu, dtype=torch.float64, generator=gen) * 0.3
    state, ab = T64(g["s0_state"]), T64(g["s0_action_buffer"])
    p = nlc.MPPIDelay(nlc.NLDynamics(build(True), 0.05), nlc.EnvCost("oderl-" + env), d, nlc.noise_sigma(nu), K, Tt,
                      "cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone())
    p.noise_dist = _Replay(raw.clone())
    act = p.command(state, ab)
    ref = omppi.mppi_command(U0.clone(), state, ab, raw, ornn.make_dynamics_rnn(sd), oenvs.RUNNING_COST["oderl-" + env],
                             d, torch.inverse(nlc.noise_sigma(nu)), 1.0, A, torch.tensor(-A), torch.tensor(A))
    np.testing.assert_allclose(act.numpy(), ref["action"].numpy(), **TOL)
    np.testing.assert_allclose(p.states.numpy(), ref["states"].numpy(), **TOL)


@pytest.mark.parametrize("hidden,B,N", [(64, 4, 1000), (128, 1, 77), (160, 6, 513), (160, 4, 1)])
def test_dtrnn_forward_vs_oracle_sizes(nlc, hidden, B, N):
    from oracle import rnn_model as ornn

    d, nu = 6, 2
    sd = ornn.make_synthetic_state_dict(11, d, nu, hidden, np.linspace(0.7, 2.9, d), [2.5])
    g = torch.Generator().manual_seed(N)
    obs = torch.randn(N, d, dtype=torch.float64, generator=g) * 2
    win = (torch.rand(N, B, nu, dtype=torch.float64, generator=g) * 2 - 1) * 5
    ts = torch.rand(N, 1, dtype=torch.float64, generator=g) * 0.1 + 0.01
    ref = ornn.forward(sd, obs, win, ts)
    with torch.no_grad():
        got = build_rnn(nlc, sd, hidden)(obs.cuda(), win.cuda(), ts.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), **TOL)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_mppi_dtrnn_dynamics_vs_reference_golden(nlc, env):
    """G9: command() with the Delta-t RNN behind the harness closure vs reference MPPIDelay + reference DeltaTRNN."""
    g = np.load(f"{GOLD}/g9_dtrnn_{env}.npz")
    model = build_rnn(nlc, load_sd(g, "sd_"), int(g["H"]))
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["nx"]), int(g["nu"]), float(g["A"])

    def make(U0):
        return nlc.MPPIDelay(
            nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-" + env), d, nlc.noise_sigma(nu),
            num_samples=K, horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A),
            u_scale=A, U_init=U0,
        )

    check_command_steps(nlc, g, make)

    # the generic path (arbitrary closures calling the HIP model per horizon step) gives the same numbers
    dyn, cost = nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-" + env)

    def make_generic(U0):
        return nlc.MPPIDelay(
            lambda s, a: dyn(s, a), lambda s, u: cost(s, u), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
            device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0,
        )

    with torch.no_grad():
        check_command_steps(nlc, g, make_generic)


def test_mppi_dtrnn_full_horizon_vs_oracle(nlc):
    """K = 4096, T = 40, 5-row action buffer, device Philox noise replayed through the CPU oracle."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl
    from oracle import rnn_model as ornn

    env, K, Tt, B = "oderl-cartpole", 4096, 40, 5
    st = onl.ENV_STATS[env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = ornn.make_synthetic_state_dict(5, d, nu, 160, st["state_std"], [A / 2.0])
    model = build_rnn(nlc, sd, 160)
    mppi = nlc.MPPIDelay(
        nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), num_samples=K, horizon=Tt,
        device="cuda", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox",
        seed=9,
    )
    state = oenvs.initial_state(env, seed=1)
    ab = (torch.rand(B, nu, dtype=torch.float64) - 0.5) * A
    U0 = mppi.U.cpu().clone()
    action = mppi.command(state.numpy(), ab)
    # replay on the CPU: bounding is idempotent, so the bounded noise the device drew serves as the raw draw
    out = omppi.mppi_command(
        U0, state, ab, mppi.noise.cpu(), ornn.make_dynamics(sd), oenvs.RUNNING_COST[env], d,
        torch.inverse(nlc.noise_sigma(nu)), 1.0, A, torch.tensor(-A), torch.tensor(A),
    )
    np.testing.assert_allclose(action.cpu().numpy(), out["action"].numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(mppi.cost_total.cpu().numpy(), out["cost_total"].numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(mppi.states.cpu().numpy(), out["states"].numpy(), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_env_step_vs_reference_env_golden(nlc, env):
    """G10: nlc_env_step / nlc_env_obs vs the REAL env classes (torch_rhs Euler step, torch_transform_states,
    diff_reward), the harness's get_action delay buffer, and the env's reset stream."""
    g = np.load(f"{GOLD}/g10_env_{env}.npz")
    name = "oderl-" + env
    tol = dict(rtol=1e-11, atol=1e-12)
    E, dt = int(g["E"]), float(g["dt"])
    for tag, fr in (("", False), ("fr_", True)):
        if tag + "s0" not in g.files:
            continue
        e = nlc.BatchedEnv(name, E, dt=dt, action_delay=0, action_buffer_size=1, friction=fr)
        obs0 = e.set_state_(T64(g[tag + "s0"]))
        np.testing.assert_allclose(obs0.cpu().numpy(), g[tag + "obs0"], **tol)
        obs1, rew = e.step(T64(g[tag + "a"]).cuda())
        np.testing.assert_allclose(e.state.cpu().numpy(), g[tag + "s1"], **tol)
        np.testing.assert_allclose(obs1.cpu().numpy(), g[tag + "obs1"], **tol)
        np.testing.assert_allclose(rew.cpu().numpy(), g[tag + "reward"], **tol)
    # closed-loop trace with the delay buffer: three identical envs in one batch
    delay, B = int(g["loop_delay"]), int(g["loop_B"])
    e = nlc.BatchedEnv(name, 3, dt=dt, action_delay=delay, action_buffer_size=B)
    e.set_state_(T64(g["s0"])[0].repeat(3, 1))
    e.action_buffer.zero_()
    for i, act in enumerate(T64(g["loop_actions"])):
        obs, rew = e.step(act.repeat(3, 1))  # host tensor in
        for k in range(3):
            np.testing.assert_allclose(e.state[k].cpu().numpy(), g["loop_s"][i], **tol)
            np.testing.assert_allclose(obs[k].cpu().numpy(), g["loop_obs"][i], **tol)
            np.testing.assert_allclose(float(rew[k]), float(g["loop_rew"][i]), **tol)
            np.testing.assert_allclose(e.action_buffer[k].cpu().numpy(), g["loop_ab"][i], **tol)
    # reset: env 0 of a batch seeded with s draws the stream of a reference env seeded with s
    e = nlc.BatchedEnv(name, 4, seed=5)
    np.testing.assert_allclose(e.state[0].cpu().numpy(), g["reset_seed5_state"], rtol=0, atol=0)
    np.testing.assert_allclose(e.get_obs()[0].cpu().numpy(), g["reset_seed5_obs"], **tol)
    with pytest.raises(ValueError):
        nlc.BatchedEnv(name, 2, action_delay=4, action_buffer_size=4)
    # per-episode reset: only the listed env is re-drawn (continuing ITS stream) and gets a zeroed action buffer
    e.step(torch.ones(4, int(g["nu"]), dtype=torch.float64))
    before_s, before_ab = e.state.clone(), e.action_buffer.clone()
    e.reset([2])
    keep = [0, 1, 3]
    assert torch.equal(e.state[keep], before_s[keep]) and torch.equal(e.action_buffer[keep], before_ab[keep])
    assert not torch.equal(e.state[2], before_s[2]) and float(e.action_buffer[2].abs().max()) == 0.0
    from oracle import envs as oenvs

    rs = np.random.RandomState(5 + 2)
    oenvs.env_reset(name, rs)  # the constructor's draw
    np.testing.assert_allclose(e.state[2].cpu().numpy(), oenvs.env_reset(name, rs).numpy(), rtol=0, atol=0)


def test_device_closed_loop_matches_host_stepped_loop(nlc):
    """BatchedMPPIDelay + BatchedEnv entirely on the device vs the same planner stepped through the CPU restatement of
    the env (oracle/envs.py, pinned by G10): identical actions, states and rewards over 6 control steps."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi

    env, E, K, Tt, B, delay = "oderl-acrobot", 5, 192, 8, 4, 1
    nx, nu, A = 6, 2, 5.0

    def planner():
        return nlc.BatchedMPPIDelay(
            nlc.OracleDynamics(env, 0.05, delay), nlc.EnvCost(env), nx, nlc.noise_sigma(nu), E, K, Tt, "cuda",
            lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=3,
            U_init=torch.zeros(E, Tt, nu, dtype=torch.float64))

    dev = nlc.BatchedEnv(env, E, action_delay=delay, action_buffer_size=B, seed=11)
    mp_dev, mp_host = planner(), planner()
    s = dev.state.cpu().clone()
    ab = torch.zeros(E, B, nu, dtype=torch.float64)
    obs = dev.get_obs()
    for _ in range(6):
        act = mp_dev.command(obs, dev.action_buffer)
        obs, rew = dev.step(act)
        # host-stepped twin
        act_h = mp_host.command(oenvs.env_obs(env, s), ab).cpu()
        np.testing.assert_allclose(act.cpu().numpy(), act_h.numpy(), rtol=1e-9, atol=1e-10)
        rews = []
        for k in range(E):
            ab[k], at = omppi.get_action(ab[k], act_h[k], delay)
            s[k], _, r = oenvs.env_step(env, s[k], at.clone(), 0.05)
            rews.append(float(r))
        np.testing.assert_allclose(dev.state.cpu().numpy(), s.numpy(), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(rew.cpu().numpy(), np.array(rews), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(dev.action_buffer.cpu().numpy(), ab.numpy(), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_node_forward_vs_reference_golden(nlc, env):
    """G11: HIP NODE.forward vs the REAL reference classes (odeint = the restated fixed-grid Euler on both sides),
    three and six Euler sub-steps."""
    g = np.load(f"{GOLD}/g11_node_{env}.npz")
    sd = load_sd(g, "sd_")
    model = build_node(nlc, sd, int(g["H"]), int(g["AUG"]))
    obs, win = T64(g["fwd_obs"]), T64(g["fwd_window"])
    with torch.no_grad():
        for tag in ("", "t2_"):
            got = model(obs.cuda(), win.cuda(), T64(g[f"fwd_{tag}ts"]).cuda()).cpu()
            np.testing.assert_allclose(got.numpy(), g[f"fwd_{tag}out"], **TOL)
        got_cpu = model(obs, win[:, -1, :], T64(g["fwd_ts"]))  # 2-D action input (train_utils.py:712-713), CPU tensors
        assert got_cpu.device.type == "cpu"
        np.testing.assert_allclose(got_cpu.numpy(), g["fwd_out"], **TOL)
    # grad mode: torch-op Euler loop on PyTorch-ROCm; output and gradients = autograd of the oracle
    from oracle import node_model as onode

    leaves = {k: (v.clone().requires_grad_() if k.startswith("x_ode_func") else v) for k, v in sd.items()}
    ref = onode.forward(leaves, obs, win, T64(g["fwd_ts"]))
    ref.square().sum().backward()
    out = model(obs.cuda(), win.cuda(), T64(g["fwd_ts"]).cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["fwd_out"], **TOL)
    out.square().sum().backward()
    for k, p_ in model.named_parameters():
        sc = float(leaves[k].grad.abs().max()) + 1e-300
        np.testing.assert_allclose(p_.grad.cpu().numpy() / sc, leaves[k].grad.numpy() / sc, rtol=1e-7, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("hidden,aug,N", [(64, 0, 500), (100, 2, 77), (128, 1, 1), (270, 1, 1030)])
def test_node_forward_vs_oracle_sizes(nlc, hidden, aug, N):
    from oracle import node_model as onode

    d, nu = 6, 2
    sd = onode.make_synthetic_state_dict(13, d, nu, hidden, aug, np.linspace(0.7, 2.9, d), [2.5])
    g = torch.Generator().manual_seed(N)
    obs = torch.randn(N, d, dtype=torch.float64, generator=g) * 2
    win = (torch.rand(N, 3, nu, dtype=torch.float64, generator=g) * 2 - 1) * 5
    ts = torch.full((N, 1), 0.07, dtype=torch.float64)
    for nt in (True, False):
        ref = onode.forward(sd, obs, win, ts, normalize=True, normalize_time=nt)
        with torch.no_grad():
            got = build_node(nlc, sd, hidden, aug, normalize_time=nt)(obs.cuda(), win.cuda(), ts.cuda()).cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), **TOL)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_mppi_node_dynamics_vs_reference_golden(nlc, env):
    """G11: command() with the NODE behind the harness closure vs reference MPPIDelay + reference NODE."""
    g = np.load(f"{GOLD}/g11_node_{env}.npz")
    model = build_node(nlc, load_sd(g, "sd_"), int(g["H"]), int(g["AUG"]))
    K, T, d, nu, A = int(g["K"]), int(g["T"]), int(g["nx"]), int(g["nu"]), float(g["A"])

    def make(U0):
        return nlc.MPPIDelay(
            nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-" + env), d, nlc.noise_sigma(nu),
            num_samples=K, horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A),
            u_scale=A, U_init=U0,
        )

    # the synthetic model is expansive (the reference ADDS the integrated normalised state to the raw state): the
    # states reach 1e3 within 8 steps and 1-ulp differences grow with them -> relative tolerance on the large entries
    check_command_steps(nlc, g, make, tol=dict(rtol=1e-8, atol=1e-8))


@pytest.mark.parametrize("kind", ["dtrnn", "node"])
def test_batched_planner_baseline_models_equal_single_planners(nlc, kind):
    """E episodes with the Delta-t RNN / NODE dynamics: K = 100 makes the 16-sample MFMA tiles straddle episodes;
    episode e is bit-identical to a single planner fed the same draws."""
    from oracle import nl_model as onl
    from oracle import node_model as onode
    from oracle import rnn_model as ornn

    env, d, nu, A = "oderl-acrobot", 6, 2, 5.0
    st = onl.ENV_STATS[env]
    if kind == "dtrnn":
        model = build_rnn(nlc, ornn.make_synthetic_state_dict(3, d, nu, 64, st["state_std"], [A / 2]), 64)
    else:
        model = build_node(nlc, onode.make_synthetic_state_dict(3, d, nu, 100, 1, st["state_std"], [A / 2]), 100, 1)
    _batched_vs_singles(nlc, lambda: nlc.NLDynamics(model, 0.05), env, E=3, K=100, T=6, n_cmd=2)


@pytest.mark.parametrize("kind", ["dtrnn", "node"])
def test_baseline_models_with_cost_callables_and_weight_updates(nlc, kind):
    """cost_external next to the baseline-model rollouts (a running_cost closure and a terminal cost keep the fused
    dynamics), and a load_state_dict between commands is picked up by the planner."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl
    from oracle import node_model as onode
    from oracle import rnn_model as ornn

    env, d, nu, A, K, Tt = "oderl-cartpole", 5, 1, 3.0, 96, 5
    st = onl.ENV_STATS[env]
    if kind == "dtrnn":
        sds = [ornn.make_synthetic_state_dict(s, d, nu, 64, st["state_std"], [A / 2]) for s in (1, 2)]
        model, dyn_of = build_rnn(nlc, sds[0], 64), ornn.make_dynamics
    else:
        sds = [onode.make_synthetic_state_dict(s, d, nu, 64, 1, st["state_std"], [A / 2]) for s in (1, 2)]
        model, dyn_of = build_node(nlc, sds[0], 64, 1), onode.make_dynamics
    cost = nlc.EnvCost(env)
    term = lambda states, actions: 0.1 * (states[:, -1, :] ** 2).sum(-1)  # noqa: E731
    g = torch.Generator().manual_seed(21)
    raws = [torch.randn(K, Tt, nu, dtype=torch.float64, generator=g) for _ in range(2)]
    U0 = torch.randn(Tt, nu, dtype=torch.float64, generator=g) * 0.3
    state, ab = _state(nlc, env, 2), (torch.rand(4, nu, dtype=torch.float64, generator=g) - 0.5) * A
    with torch.no_grad():
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), lambda s, u: cost(s, u), d, nlc.noise_sigma(nu), K, Tt, "cpu",
                          lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone(),
                          terminal_state_cost=term)
        assert p.cost_external
        p.noise_dist = _Replay(*[r.clone() for r in raws])
        U = U0.clone()
        for i, sd in enumerate(sds):
            if i:
                model.load_state_dict(sd)
            act = p.command(state, ab)
            ref = omppi.mppi_command(U, state, ab, raws[i], dyn_of(sd), oenvs.RUNNING_COST[env], d,
                                     torch.inverse(nlc.noise_sigma(nu)), 1.0, A, torch.tensor(-A), torch.tensor(A),
                                     terminal_state_cost=term)
            np.testing.assert_allclose(act.numpy(), ref["action"].numpy(), rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(p.cost_total.numpy(), ref["cost_total"].numpy(), rtol=1e-8, atol=1e-8)
            U = ref["U"].clone()


@pytest.mark.parametrize("kind,K,Tt,B", [("dtrnn", 17, 1, 1), ("dtrnn", 130, 3, 6), ("node", 17, 1, 1), ("node", 130, 3, 6),
                                         ("rnn", 33, 2, 2)])
def test_baseline_planners_edge_shapes_vs_oracle(nlc, kind, K, Tt, B):
    """Ragged K (below / across one 16-sample tile), T = 1, one-row and six-row action buffers, nu = 2."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl
    from oracle import node_model as onode
    from oracle import rnn_model as ornn

    env, d, nu, A = "oderl-acrobot", 6, 2, 5.0
    st = onl.ENV_STATS[env]
    if kind == "dtrnn":
        sd = ornn.make_synthetic_state_dict(8, d, nu, 128, st["state_std"], [A / 2])
        model, dyn = build_rnn(nlc, sd, 128), ornn.make_dynamics(sd)
    elif kind == "rnn":
        sd = ornn.make_synthetic_state_dict(8, d, nu, 64, st["state_std"], [A / 2], time_input=False)
        model = nlc.RNN(d, nu, hidden_units=64, state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0]),
                        action_std=np.array([1.0]), normalize=True).double()
        model.load_state_dict({k: v for k, v in sd.items() if k != "dt"})
        model, dyn = model.cuda(), ornn.make_dynamics_rnn(sd)
    else:
        sd = onode.make_synthetic_state_dict(8, d, nu, 128, 1, st["state_std"], [A / 2])
        model, dyn = build_node(nlc, sd, 128, 1), onode.make_dynamics(sd)
    g = torch.Generator().manual_seed(K * 7 + B)
    raw = torch.randn(K, Tt, nu, dtype=torch.float64, generator=g) @ torch.linalg.cholesky(nlc.noise_sigma(nu)).T
    U0 = torch.randn(Tt, nu, dtype=torch.float64, generator=g) * 0.3
    state = _state(nlc, env, 3)
    ab = (torch.rand(B, nu, dtype=torch.float64, generator=g) - 0.5) * A
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, Tt, "cpu", lambda_=1.0,
                      u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone())
    p.noise_dist = _Replay(raw.clone())
    act = p.command(state, ab)
    ref = omppi.mppi_command(U0.clone(), state, ab, raw, dyn, oenvs.RUNNING_COST[env], d,
                             torch.inverse(nlc.noise_sigma(nu)), 1.0, A, torch.tensor(-A), torch.tensor(A))
    np.testing.assert_allclose(act.numpy(), ref["action"].numpy(), **TOL)
    np.testing.assert_allclose(p.cost_total.numpy(), ref["cost_total"].numpy(), **TOL)
    np.testing.assert_allclose(p.states.numpy(), ref["states"].numpy(), **TOL)
