"""Ad-hoc GPU diagnostics (not a test): prints stage-wise errors of the HIP path vs the CPU oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
from oracle import ilt as oilt, nl_model as onl, envs as oenvs, mppi as omppi
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_common import build_model, load_sd, T64, GOLD
torch.set_printoptions(precision=6, linewidth=200)

def err(a, b):
    return float((a - b).abs().max())

print("==== GRU mode 0")
for env in ("cartpole", "acrobot"):
    g2 = np.load(f"{GOLD}/g2_stages_{env}.npz"); sd = load_sd(g2); model = build_model(nlc, sd)
    g3 = np.load(f"{GOLD}/g3_nl_{env}.npz")
    for name, win_norm in (("g2_in", T64(g2["gru_in"])), ("g3_win_norm", T64(g3["fwd_window"]) / sd["action_std"]),
                           ("zeros", torch.zeros(48, 4, win_shape := T64(g2["gru_in"]).shape[-1], dtype=torch.float64)),
                           ("ones", torch.ones(20, 4, T64(g2["gru_in"]).shape[-1], dtype=torch.float64))):
        ref = onl.gru_encoder(sd, win_norm)
        with torch.no_grad():
            got = model.encode_actions((win_norm * sd["action_std"] + sd["action_mean"]).cuda()).cpu()
        e = (got - ref).abs().amax(dim=1)
        print(env, name, "max err", float(e.max()), "bad rows", torch.nonzero(e > 1e-9).flatten().tolist()[:40], "N", len(e))
        if e.max() > 1e-9:
            print("  got", got[:3].tolist(), "ref", ref[:3].tolist())

print("==== model forward per env")
for env in ("cartpole", "pendulum", "acrobot"):
    g3 = np.load(f"{GOLD}/g3_nl_{env}.npz"); sd = load_sd(g3); model = build_model(nlc, sd)
    obs, win, ts = T64(g3["fwd_obs"]), T64(g3["fwd_window"]), T64(g3["fwd_ts"])
    with torch.no_grad():
        got = model(obs.cuda(), win.cuda(), ts.cuda()).cpu()
    ref = T64(g3["fwd_out"])
    print(env, "fwd err", err(got, ref), "nt3?", "got[0]", got[0].tolist(), "ref[0]", ref[0].tolist())
    # staged through the python path with the fused GRU
    with torch.no_grad():
        pa = model.encode_actions(win.cuda()).cpu()
    a = (win - sd["action_mean"]) / sd["action_std"]
    print("   gru err", err(pa, onl.gru_encoder(sd, a)))

print("==== ILT fourier")
for d, S in ((5, 17), (3, 17), (5, 32), (2, 17), (1, 17), (4, 17)):
    torch.manual_seed(d * 100 + S)
    N = 1537
    theta = (torch.rand(N, d, S, dtype=torch.float64) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64) * 2 - 1) * np.pi / 2 * 0.999
    t = torch.rand(N, dtype=torch.float64) * 2 + 0.05
    for opts in (None, dict(scale=3.0, alpha=1e-2)):
        ref = oilt.ilt_from_sphere(theta, phi, t, "fourier", opts).reshape(-1)
        got = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), "fourier", opts).cpu().reshape(-1)
        e = ((got - ref).abs() / ref.abs().max())
        bad = torch.nonzero(e > 1e-9).flatten()
        print(d, S, opts, "max rel err", float(e.max()), "nbad", len(bad), "first bad rows", bad[:8].tolist(), "last", bad[-4:].tolist())

print("==== dehoog")
for S in (33, 17, 9):
    torch.manual_seed(S)
    N, d = 700, 5
    t = torch.rand(N, dtype=torch.float64) * 2 + 0.05
    alpha, tol, scale = oilt.ilt_options("dehoog")
    sr, si, _, _ = oilt.query_points(t, S, alpha, tol, scale)
    s = torch.complex(sr, si).unsqueeze(1)
    a = (torch.rand(N, d, 1, dtype=torch.float64) + 0.5); w = (torch.rand(N, d, 1, dtype=torch.float64) * 3 + 0.5)
    F = (s + a) / ((s + a) ** 2 + w**2)
    theta, phi = oilt.complex_to_sphere(F.real, F.imag)
    ref = oilt.ilt_from_sphere(theta, phi, t, "dehoog")
    got = nlc.ilt_reconstruct(theta.cuda(), phi.cuda(), t.cuda(), "dehoog").cpu()
    exact = torch.exp(-a.squeeze(-1) * t.view(-1, 1)) * torch.cos(w.squeeze(-1) * t.view(-1, 1))
    print(S, "got-ref", err(got, ref), "ref-exact", err(ref, exact), "got-exact", err(got, exact), got[0].tolist(), ref[0].tolist())

print("==== full-size U consistency")
env, K, T, A, d, nu = "oderl-cartpole", 16384, 40, 3.0, 5, 1
st = onl.ENV_STATS[env]
sd = onl.make_synthetic_state_dict(0, d, nu, 128, 17, st["state_std"], [A / 2], tame=True)
model = build_model(nlc, sd)
for dev in ("cpu", "cuda"):
    torch.manual_seed(0)
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), 256, T, dev, lambda_=1.0,
                         u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A)
    Ub = mppi.U.cpu().clone()
    mppi.command(nlc.initial_state(env), torch.zeros(4, nu, dtype=torch.float64))
    V, eps = mppi.perturbed_action.cpu(), mppi.noise.cpu()
    Ush = torch.roll(Ub, -1, 0); Ush[-1] = 0
    print(dev, "U+eps-V", err(Ush + eps, V), "Ub[:3]", Ub[:3].flatten().tolist(), "(V-eps)[0,:3]", (V - eps)[0, :3].flatten().tolist())
