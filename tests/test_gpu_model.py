"""GPU parity tests (run with ``-m gpu`` on an MI355X), through the C ABI of libnlc_hip.so via the drop-in Python mirror:
the model stages (a6-a8): GRU encoder, representation function, NeuralLaplaceModel.forward, training through the HIP ILT, other widths.  Helpers and tolerances: tests/gpu_common.py.
"""

import glob
import os

import numpy as np
import pytest
import torch

from gpu_common import *  # noqa: F401,F403
from gpu_common import GOLD, TOL, T64, load_sd, build_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_gru_encoder_vs_reference_golden(nlc, env, encoder_mode):
    """G2: HIP GRU encoder vs the REAL reference ReverseGRUEncoder (nn.GRU) outputs."""
    g = np.load(f"{GOLD}/g2_stages_{env}.npz")
    sd = load_sd(g)
    model = build_model(nlc, sd)
    # G2 fed already-normalised windows; un-normalise so the kernel's (x - mean)/std reproduces them
    win = T64(g["gru_in"]) * sd["action_std"] + sd["action_mean"]
    with torch.no_grad():
        got = model.encode_actions(win.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), g["gru_out"], **TOL)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_repfunc_kernel_vs_reference_golden(nlc, env):
    """G2, row a8 pinned directly on the GPU: the MFMA representation-function kernel (``nl_repfunc_kernel`` behind
    ``nlc_rep_func``) on the fixture's random input rows vs the outputs of the REAL reference module
    ``LaplaceRepresentationFunc.forward`` (w_nl.py:55-63): theta = pi tanh(.), phi = (pi/2) tanh(.) per (dim, term)."""
    g = np.load(f"{GOLD}/g2_stages_{env}.npz")
    model = build_model(nlc, load_sd(g))
    rep_in = T64(g["rep_in"])
    with torch.no_grad():
        theta, phi = model.rep_func_hip(rep_in.cuda())
    assert theta.shape == g["rep_theta"].shape and phi.shape == g["rep_phi"].shape
    np.testing.assert_allclose(theta.cpu().numpy(), g["rep_theta"], **TOL)
    np.testing.assert_allclose(phi.cpu().numpy(), g["rep_phi"], **TOL)
    # ragged N (not a multiple of the 16-row MFMA tile) and a single row
    with torch.no_grad():
        th1, ph1 = model.rep_func_hip(rep_in[:1].cuda())
    np.testing.assert_allclose(th1.cpu().numpy(), g["rep_theta"][:1], **TOL)
    np.testing.assert_allclose(ph1.cpu().numpy(), g["rep_phi"][:1], **TOL)


@pytest.mark.parametrize("env", ["cartpole", "pendulum", "acrobot"])
def test_model_forward_vs_golden(nlc, env, encoder_mode):
    """G3: fused HIP NeuralLaplaceModel.forward vs the reference model (ILT body = build's restatement)."""
    g = np.load(f"{GOLD}/g3_nl_{env}.npz")
    model = build_model(nlc, load_sd(g))
    with torch.no_grad():
        got = model(T64(g["fwd_obs"]).cuda(), T64(g["fwd_window"]).cuda(), T64(g["fwd_ts"]).cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), g["fwd_out"], **TOL)
    # CPU-resident inputs (the reference's default device) give the same numbers, returned on the CPU
    with torch.no_grad():
        got_cpu = model(T64(g["fwd_obs"]), T64(g["fwd_window"]), T64(g["fwd_ts"]))
    assert got_cpu.device.type == "cpu"
    np.testing.assert_allclose(got_cpu.numpy(), g["fwd_out"], **TOL)


@pytest.mark.parametrize("env", ["cartpole", "acrobot"])
def test_model_forward_dehoog(nlc, env):
    g = np.load(f"{GOLD}/g3_nl_{env}.npz")
    model = build_model(nlc, load_sd(g, "w33::"), S=33, algo="dehoog")
    with torch.no_grad():
        got = model(T64(g["fwd_obs"]).cuda(), T64(g["fwd_window"]).cuda(), T64(g["fwd_ts"]).cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), g["fwd33_out"], rtol=1e-6, atol=1e-6)


def test_model_forward_general_t_and_ragged(nlc):
    """Per-row prediction times (not the planner's constant dt), N not a multiple of the wave tile."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS["oderl-acrobot"]
    sd = onl.make_synthetic_state_dict(3, st["d"], st["nu"], 128, 17, st["state_std"], [st["act_high"] / 2], tame=True)
    model = build_model(nlc, sd)
    torch.manual_seed(9)
    for N in (1, 15, 16, 17, 129):
        obs = torch.randn(N, st["d"], dtype=torch.float64)
        win = torch.randn(N, 5, st["nu"], dtype=torch.float64) * 2  # B = 5 window
        ts = torch.rand(N, 1, dtype=torch.float64) * 0.2 + 0.01
        ref = onl.nl_forward(sd, obs, win, ts, S=17).reshape(N, -1)
        with torch.no_grad():
            got = model(obs.cuda(), win.cuda(), ts.cuda()).cpu().reshape(N, -1)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), **TOL)


def test_model_requires_no_grad_and_double(nlc):
    from oracle import nl_model as onl

    st = onl.ENV_STATS["oderl-cartpole"]
    sd = onl.make_synthetic_state_dict(0, 5, 1, 128, 17, st["state_std"], [1.5])
    model = build_model(nlc, sd)
    for p_ in model.parameters():
        p_.requires_grad_(False)
    with pytest.raises(NotImplementedError):  # grad mode with nothing to train: the fused path is inference-only
        model(torch.zeros(2, 5).double().cuda(), torch.zeros(2, 4, 1).double().cuda(), torch.ones(2, 1).double().cuda())
    with torch.no_grad(), pytest.raises(NotImplementedError):
        model.float()(torch.zeros(2, 5).cuda(), torch.zeros(2, 4, 1).cuda(), torch.ones(2, 1).cuda())


@pytest.mark.parametrize("env", ["cartpole", "acrobot"])
def test_model_trains_through_hip_ilt(nlc, env):
    """Grad-mode forward (train_utils.py:388-407 trains through model(...)): GRU / MLP on PyTorch-ROCm, line integral
    forward AND backward in HIP.  Output and every parameter gradient equal autograd through the CPU restatement."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS["oderl-" + env]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(3, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    g = torch.Generator().manual_seed(17)
    N = 203
    obs = torch.randn(N, d, dtype=torch.float64, generator=g) * torch.tensor(st["state_std"])
    win = (torch.rand(N, 4, nu, dtype=torch.float64, generator=g) * 2 - 1) * A
    ts = torch.rand(N, 1, dtype=torch.float64, generator=g) * 0.08 + 0.02
    target = torch.randn(N, d, dtype=torch.float64, generator=g)
    # oracle side: the state_dict tensors as leaves
    names = [k for k in sd if k.startswith(("action_encoder.", "laplace_rep_func."))]
    leaves = {k: (v.clone().requires_grad_() if k in names else v) for k, v in sd.items()}
    ref = onl.nl_forward(leaves, obs, win, ts, S=17)
    ((ref - target) ** 2).mean().backward()
    model = build_model(nlc, sd)
    model.train()
    got = model(obs.cuda(), win.cuda(), ts.cuda())
    assert got.requires_grad
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-8, atol=1e-10)
    ((got - target.cuda()) ** 2).mean().backward()
    for k, p_ in model.named_parameters():
        ref_g = leaves[k].grad
        sc = float(ref_g.abs().max()) + 1e-300
        np.testing.assert_allclose(p_.grad.cpu().numpy() / sc, ref_g.numpy() / sc, rtol=1e-7, atol=1e-9, err_msg=k)
    # one optimiser step changes the weights; the planner's fused (inference) path picks them up
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    opt.step()
    with torch.no_grad():
        after = model(obs.cuda(), win.cuda(), ts.cuda())
    assert not torch.allclose(after, got.detach())
    with torch.no_grad():
        twin = build_model(nlc, {k: v.detach().cpu() for k, v in model.state_dict().items()})(obs.cuda(), win.cuda(), ts.cuda())
    np.testing.assert_allclose(after.cpu().numpy(), twin.cpu().numpy(), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("algo,S", [("fixed_tablot", 11), ("stehfest", 8), ("dehoog", 9)])
def test_model_with_linear_ilt_trains_through_hip_ilt(nlc, monkeypatch, algo, S):
    """The same for a model configured with fixed_tablot / stehfest / dehoog (the reference trains through whichever
    ilt_algorithm its config names, train_utils.py:388-407): forward and every parameter gradient vs the restatement."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS["oderl-pendulum"]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(5, d, nu, 128, S, st["state_std"], [A / 2], tame="dehoog" if algo == "dehoog" else True)
    g = torch.Generator().manual_seed(23)
    N = 77
    obs = torch.randn(N, d, dtype=torch.float64, generator=g) * torch.tensor(st["state_std"])
    win = (torch.rand(N, 4, nu, dtype=torch.float64, generator=g) * 2 - 1) * A
    ts = torch.rand(N, 1, dtype=torch.float64, generator=g) * 0.08 + 0.02
    target = torch.randn(N, d, dtype=torch.float64, generator=g)
    names = [k for k in sd if k.startswith(("action_encoder.", "laplace_rep_func."))]
    leaves = {k: (v.clone().requires_grad_() if k in names else v) for k, v in sd.items()}
    if algo == "dehoog":  # the oracle's de Hoog writes its table in place: differentiate the functional twin instead
        from oracle import ilt as oilt

        plain = onl.nl_forward(sd, obs, win, ts, S=S, ilt_algorithm=algo)
        monkeypatch.setitem(oilt.LINE_INTEGRATE, "dehoog", dehoog_line_integrate_functional)
        np.testing.assert_allclose(onl.nl_forward(sd, obs, win, ts, S=S, ilt_algorithm=algo).numpy(), plain.numpy(), rtol=1e-9, atol=1e-11)
    ref = onl.nl_forward(leaves, obs, win, ts, S=S, ilt_algorithm=algo)
    ((ref - target) ** 2).mean().backward()
    model = build_model(nlc, sd, S=S, algo=algo)
    model.train()
    got = model(obs.cuda(), win.cuda(), ts.cuda())
    assert got.requires_grad
    sc = float(ref.detach().abs().max())
    np.testing.assert_allclose(got.detach().cpu().numpy() / sc, ref.detach().numpy().reshape(got.shape) / sc, rtol=1e-7, atol=1e-9)
    ((got - target.cuda().reshape(got.shape)) ** 2).mean().backward()
    for k, p_ in model.named_parameters():
        ref_g = leaves[k].grad
        sc = float(ref_g.abs().max()) + 1e-300
        np.testing.assert_allclose(p_.grad.cpu().numpy() / sc, ref_g.numpy() / sc, rtol=1e-6, atol=1e-8, err_msg=k)


@pytest.mark.parametrize("h,S,algo", [(64, 33, "fourier"), (64, 17, "dehoog"), (256, 17, "fourier"), (256, 21, "dehoog"), (64, 5, "dehoog")])
def test_other_hidden_widths_forward_and_planner(nlc, h, S, algo):
    """hidden_units = 64 (the class default, w_nl.py:72, with its default 33 terms) and 256: model.forward and one planning
    step on every rollout body that exists for the width, against the oracle."""
    from oracle import envs as oenvs
    from oracle import mppi as omppi
    from oracle import nl_model as onl

    env, K, T, A = "oderl-pendulum", 200, 7, 2.0
    st = onl.ENV_STATS[env]
    d, nu = st["d"], st["nu"]
    sd = onl.make_synthetic_state_dict(5, d, nu, h, S, st["state_std"], [A / 2], tame="dehoog" if algo == "dehoog" else True)
    model = build_model(nlc, sd, S=S, algo=algo)
    assert model.hidden_units == h
    torch.manual_seed(h + S)
    obs, win = torch.randn(37, d, dtype=torch.float64), torch.randn(37, 4, nu, dtype=torch.float64)
    ts = torch.rand(37, 1, dtype=torch.float64) * 0.1 + 0.02
    ref = onl.nl_forward(sd, obs, win, ts, S=S, ilt_algorithm=algo)
    with torch.no_grad():
        got = model(obs.cuda(), win.cuda(), ts.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy().reshape(got.shape), rtol=1e-8, atol=1e-9)
    raw = torch.randn(K, T, nu, dtype=torch.float64)
    U0 = torch.randn(T, nu, dtype=torch.float64) * 0.2
    state, ab = nlc.initial_state(env), torch.randn(4, nu, dtype=torch.float64) * 0.3
    sig = nlc.noise_sigma(nu)
    tsk = torch.full((K, 1), 0.05, dtype=torch.float64)
    ref = omppi.mppi_command(U0.clone(), state, ab, raw.clone(), onl.nl_dynamics(sd, tsk, S=S, ilt_algorithm=algo),
                             oenvs.RUNNING_COST[env], d, torch.inverse(sig), 1.0, A, torch.tensor(-A), torch.tensor(A))
    for variant in ((1, 2, 3) if algo == "fourier" else (0,)):  # 3: the fused one-launch body exists for every width
        mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, sig, K, T, "cpu", lambda_=1.0,
                             u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, U_init=U0.clone(),
                             planner_options={"rollout_variant": variant})
        mppi.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
        with torch.no_grad():
            act = mppi.command(state, ab)
        np.testing.assert_allclose(mppi.states.numpy(), ref["states"].numpy(), rtol=1e-7, atol=1e-8, err_msg=f"variant {variant}")
        np.testing.assert_allclose(act.numpy(), ref["action"].numpy(), rtol=1e-7, atol=1e-8)


@pytest.mark.no_sliced_instance
@pytest.mark.parametrize("name", ["h64_pendulum", "h256_acrobot"])
def test_other_hidden_widths_vs_reference_golden(nlc, name, encoder_mode):
    """G14: hidden_units 64 (class default with its 33 terms) and 256 against the REAL reference classes: HIP GRU encoder
    (g = 32 / 128) vs nn.GRU, the representation kernel vs the module, model.forward, and two commands of the reference
    planner on every rollout body of the width."""
    g = np.load(f"{GOLD}/g14_width_{name}.npz")
    sd = load_sd(g)
    env = "oderl-" + name.split("_")[1]
    d, nu, S, K, T, A = int(g["d"]), int(g["nu"]), int(g["S"]), int(g["K"]), int(g["T"]), float(g["A"])
    raw = {k: v.clone() for k, v in sd.items()}
    raw["laplace_rep_func.linear_tanh_stack.4.bias"][d * S :] += 3.0  # stage fixtures predate the phi-bias shift (-3)
    m_raw = build_model(nlc, raw, S=S)
    with torch.no_grad():
        win = T64(g["gru_in"]) * raw["action_std"] + raw["action_mean"]
        np.testing.assert_allclose(m_raw.encode_actions(win.cuda()).cpu().numpy(), g["gru_out"], **TOL)
        th, ph = m_raw.rep_func_hip(T64(g["rep_in"]).cuda())
        np.testing.assert_allclose(th.cpu().numpy(), g["rep_theta"], **TOL)
        np.testing.assert_allclose(ph.cpu().numpy(), g["rep_phi"], **TOL)
        model = build_model(nlc, sd, S=S)
        got = model(T64(g["fwd_obs"]).cuda(), T64(g["fwd_window"]).cuda(), T64(g["fwd_ts"]).cuda()).cpu()
        np.testing.assert_allclose(got.numpy(), g["fwd_out"], **TOL)
        for variant in (1, 2, 3):
            def make(U0, variant=variant):
                return nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), num_samples=K,
                                     horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A),
                                     u_scale=A, U_init=U0, planner_options={"rollout_variant": variant})

            check_command_steps(nlc, g, make)


def test_model_with_linear_ilt_and_cme_constructor(nlc):
    """A NeuralLaplaceModel configured with fixed_tablot runs (HIP GRU -> torch rep func -> HIP ILT) and plans on the
    staged all-HIP path (representation kernel -> slot-major linear ILT -> state kernel per horizon step); with "cme" the constructor snaps the term count like the reference (w_nl.py:86-88) and the forward
    says why the method cannot run here."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS["oderl-pendulum"]
    d, nu, S = st["d"], st["nu"], 17
    sd = onl.make_synthetic_state_dict(3, d, nu, 128, S, st["state_std"], [1.0], tame=True)
    model = build_model(nlc, sd, S=S, algo="fixed_tablot")
    torch.manual_seed(2)
    obs, win = torch.randn(21, d, dtype=torch.float64), torch.randn(21, 4, nu, dtype=torch.float64)
    ts = torch.full((21, 1), 0.05, dtype=torch.float64)
    ref = onl.nl_forward(sd, obs, win, ts, S=S, ilt_algorithm="fixed_tablot")
    with torch.no_grad():
        got = model(obs.cuda(), win.cuda(), ts.cuda()).cpu()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got.numpy(), ref.numpy().reshape(got.shape), rtol=1e-7, atol=1e-9 * scale)
    # planner: the staged all-HIP path (round 3) against the generic path, whose dynamics callable is the model's forward
    K, T = 200, 5
    raw = torch.randn(K, T, nu, dtype=torch.float64) * 0.5
    state, ab = nlc.initial_state("oderl-pendulum"), torch.randn(4, nu, dtype=torch.float64) * 0.3
    out = {}
    dyn_obj = nlc.NLDynamics(model, 0.05)
    for name, dyn in (("staged", dyn_obj), ("generic", lambda s_, w_: dyn_obj(s_, w_))):
        mppi = nlc.MPPIDelay(dyn, nlc.EnvCost("oderl-pendulum"), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                             u_min=torch.tensor(-2.0), u_max=torch.tensor(2.0), u_scale=2.0,
                             U_init=torch.zeros(T, nu, dtype=torch.float64), planner_options={"recognise_closures": 0})
        assert mppi.fused == (name == "staged")
        mppi.noise_dist = type("R", (), {"sample": staticmethod(lambda shape: raw)})()
        with torch.no_grad():
            act = mppi.command(state, ab)
        out[name] = (mppi.states.clone(), mppi.cost_total.clone(), act.clone())
    for a, b in zip(out["staged"], out["generic"]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-6, atol=1e-8)
    cme = nlc.NeuralLaplaceModel(d, nu, d, hidden_units=128, s_recon_terms=33, ilt_algorithm="cme", state_mean=np.zeros(d),
                                 state_std=np.ones(d), action_mean=np.array([0]), action_std=np.array([1.0]))
    assert cme.s_recon_terms == 31 and cme.laplace_rep_func.linear_tanh_stack[4].out_features == 2 * d * 31
    with pytest.raises(NotImplementedError, match="cme"):
        with torch.no_grad():
            cme.double().cuda()(obs.cuda(), win.cuda(), ts.cuda())


def test_dehoog_model_forward_multi_time_uses_torch_repfunc(nlc):
    """Several time points per row (Tt > 1) go through laplace_reconstruct with the torch rep-func: same numbers."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS["oderl-pendulum"]
    sd = onl.make_synthetic_state_dict(4, 3, 1, 128, 17, st["state_std"], [1.0], tame=True)
    for algo in ("fourier", "dehoog"):
        model = build_model(nlc, sd, S=17, algo=algo)
        torch.manual_seed(1)
        obs, win = torch.randn(9, 3, dtype=torch.float64), torch.randn(9, 4, 1, dtype=torch.float64)
        ts = torch.rand(9, 3, dtype=torch.float64) * 0.2 + 0.02
        ref = onl.nl_forward(sd, obs, win, ts, S=17, ilt_algorithm=algo)
        with torch.no_grad():
            got = model(obs.cuda(), win.cuda(), ts.cuda()).cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)


def test_model_forward_constant_time_path(nlc):
    """model.forward with ONE query time for every row (what the harness closure passes) takes the folded-bias kernel
    (nlc_model_forward_const_t); rows of a call with per-row times that happen to carry the same t go through the general
    kernel and must agree with it to rounding; a Python float works too; new weights refresh the fold."""
    from oracle import nl_model as onl

    st = onl.ENV_STATS["oderl-cartpole"]
    d, nu, A = st["d"], st["nu"], st["act_high"]
    sd = onl.make_synthetic_state_dict(21, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
    model = build_model(nlc, sd)
    torch.manual_seed(3)
    N = 777
    obs = torch.randn(N, d, dtype=torch.float64).cuda()
    win = (torch.randn(N, 4, nu, dtype=torch.float64) * 0.7).cuda()
    for rep in range(2):
        for tval in (0.05, 0.11):
            ts_const = torch.full((N, 1), tval, dtype=torch.float64, device="cuda")
            ts_mixed = ts_const.clone()
            ts_mixed[0, 0] = 0.2  # one different row: the whole call takes the general per-row kernel
            with torch.no_grad():
                a = model(obs, win, ts_const)
                assert model._const_ts_cache[1] == tval
                b = model(obs, win, ts_mixed)
                c = model(obs, win, tval)
            np.testing.assert_allclose(a[1:].cpu().numpy(), b[1:].cpu().numpy(), rtol=1e-11, atol=1e-13)
            assert torch.equal(a, c)
            ref = onl.nl_forward(sd, obs.cpu(), win.cpu(), ts_const.cpu(), S=17)
            np.testing.assert_allclose(a.cpu().numpy(), ref.numpy().reshape(a.shape), rtol=1e-9, atol=1e-10)
        # new weights: the folded bias of the constant-time path must follow
        sd = onl.make_synthetic_state_dict(22 + rep, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
        model.load_state_dict(sd)


@pytest.mark.parametrize("env", ["cartpole", "pendulum"])
def test_nl_model_with_time_channel_vs_reference_golden(nlc, env, encoder_mode):
    """G5b: encode_obs_time NL model (GRU input nu+1).  forward() on explicit (N, B, nu+1) windows, and the planner with
    the harness closure's constant time channel B-1..0 (mppi_with_model.py:110-119): fused kernel and generic path."""
    g = np.load(f"{GOLD}/g5_nl_obs_time_{env}.npz")
    sd = load_sd(g)
    d, nu, K, T, A = int(g["d"]), int(g["nu"]), int(g["K"]), int(g["T"]), float(g["A"])
    m = nlc.NeuralLaplaceModel(
        d, nu, d, hidden_units=128, s_recon_terms=17, ilt_algorithm="fourier", encode_obs_time=True,
        state_mean=np.zeros(d), state_std=np.ones(d), action_mean=np.array([0] * nu), action_std=np.array([1.0]),
        normalize=True, normalize_time=True,
    ).double()
    m.load_state_dict(sd)
    m = m.cuda()
    with torch.no_grad():
        out = m(T64(g["fwd_obs"]).cuda(), T64(g["fwd_window"]).cuda(), T64(g["fwd_ts"]).cuda())
        np.testing.assert_allclose(out.cpu().numpy(), g["fwd_out"], **TOL)

        def make(U0, fused=True):
            dyn = nlc.NLDynamics(m, 0.05)
            p = nlc.MPPIDelay(
                dyn if fused else (lambda s, w: dyn(s, w)), nlc.EnvCost("oderl-" + env), d, nlc.noise_sigma(nu),
                num_samples=K, horizon=T, device="cpu", lambda_=1.0, u_min=torch.tensor(-A), u_max=torch.tensor(A),
                u_scale=A, U_init=U0,
            )
            assert p.fused == fused
            return p

        check_command_steps(nlc, g, make)
        check_command_steps(nlc, g, lambda U0: make(U0, fused=False))


def test_shapes_without_an_mfma_instance_run_the_reference_op_sequence_on_the_gpu(nlc):
    """VERDICT r4 item 5 / SURVEY 8b ("falls back to the torch path on UNSUPPORTED"): the reference constructors take any
    hidden_units / state_dim (w_nl.py:67-83); the MFMA kernels exist for hidden_units 64 / 128 / 256 and state_dim <= 6.
    hidden_units = 96 with state_dim = 7 must not raise: model.forward runs the op sequence of w_nl.py:117-145 with the GRU /
    MLP on PyTorch-ROCm and the contour / sphere map / line integral in the HIP ILT kernels (one warning), and MPPIDelay plans
    on its callables path (sampling, bounding, weights, U update still the HIP kernels) -- both against the oracle at 1e-9."""
    import warnings

    from oracle import mppi as omppi
    from oracle import nl_model as onl

    d, nu, h, S, K, T, A = 7, 1, 96, 17, 300, 8, 2.0
    sd = onl.make_synthetic_state_dict(4, d, nu, h, S, [1.0 + 0.3 * i for i in range(d)], [A / 2], tame=True)
    for where in ("cuda", "cpu"):  # weights on the GPU, and a host-resident model (device copy of the sub-modules)
        model = build_model(nlc, sd, S=S, device=where)
        torch.manual_seed(5)
        N = 37
        obs = torch.randn(N, d, dtype=torch.float64)
        win = torch.randn(N, 4, nu, dtype=torch.float64)
        ts = torch.full((N, 1), 0.05, dtype=torch.float64)
        ref = onl.nl_forward(sd, obs, win, ts, S=S).reshape(N, d)
        with warnings.catch_warnings(record=True) as seen, torch.no_grad():
            warnings.simplefilter("always")
            got = model(obs.cuda(), win.cuda(), ts.cuda()).cpu().reshape(N, d)
            model(obs.cuda(), win.cuda(), ts.cuda())
        assert sum("hidden_units must be 64, 128 or 256" in str(w.message) for w in seen) == 1, [str(w.message) for w in seen]
        assert "hidden_units" in model.hip_unsupported()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), **TOL)
    # the planner: a 7-dim state has no env cost kernel -- the running cost is the caller's callable, as in the reference
    model = build_model(nlc, sd, S=S)
    cost = lambda x, u: (x * x).sum(-1) + 0.01 * (u * u).sum(-1)  # noqa: E731
    torch.manual_seed(6)
    raw = torch.randn(K, T, nu, dtype=torch.float64)
    U0 = torch.randn(T, nu, dtype=torch.float64) * 0.1
    state, ab = torch.randn(d, dtype=torch.float64) * 0.1, torch.zeros(4, nu, dtype=torch.float64)
    with warnings.catch_warnings(record=True) as seen, torch.no_grad():
        warnings.simplefilter("always")
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), cost, d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0, u_min=torch.tensor(-A),
                          u_max=torch.tensor(A), u_scale=A, U_init=U0.clone())
        p.noise_dist = _Replay(raw)
        action = p.command(state, ab)
    assert p.unsupported_shape and p.fused_dynamics is False and p.rollout_body == "callables"
    assert any("planning on the callables path" in str(w.message) for w in seen)
    ts = torch.full((K, 1), 0.05, dtype=torch.float64)
    ref = omppi.mppi_command(U0.clone(), state, ab, raw.clone(), onl.nl_dynamics(sd, ts, S=S), cost, d,
                             torch.inverse(nlc.noise_sigma(nu)), 1.0, A, torch.tensor(-A), torch.tensor(A))
    np.testing.assert_allclose(action.numpy(), ref["action"].numpy(), **TOL)
    np.testing.assert_allclose(p.cost_total.numpy(), ref["cost_total"].numpy(), **TOL)
    np.testing.assert_allclose(p.states.numpy(), ref["states"].numpy(), **TOL)
    np.testing.assert_allclose(p.U.numpy(), ref["U"].numpy(), **TOL)
    # a state_dim beyond the descriptor (9 > NLC_MAX_D) is the same story, not an IndexError
    sd9 = onl.make_synthetic_state_dict(4, 9, nu, 64, 9, [1.0] * 9, [A / 2], tame=True)
    m9 = build_model(nlc, sd9, S=9)
    obs9 = torch.randn(5, 9, dtype=torch.float64)
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        got9 = m9(obs9.cuda(), win[:5].cuda(), ts[:5].cuda()).cpu()
    np.testing.assert_allclose(got9.numpy(), onl.nl_forward(sd9, obs9, win[:5], ts[:5], S=9).numpy(), **TOL)
