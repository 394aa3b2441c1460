"""BASELINE configs[4]: de Hoog (33 terms) vs Fourier ILT ablation on one MI355X.

Stand-alone ILT kernels at N = K*T = 655360 points (d = 5): time, algorithmic GB/s, fraction of HBM peak; and the
planner (K=16384, T=40, cartpole) with a Fourier model (fused path) vs a de Hoog S=33 model (staged/generic path).
Writes one JSON document to stdout."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
from neurallaplacecontrol_amd.laplace import default_ctx

N, d = 16384 * 40, 5
out = {"points": N, "d": d, "kernels": []}
ctx = default_ctx(0)
g = torch.Generator(device="cuda").manual_seed(1)
for algo, S in (("fourier", 17), ("fourier", 33), ("dehoog", 17), ("dehoog", 33)):
    theta = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2) * 0.98
    t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
    for _ in range(2):
        nlc.ilt_reconstruct(theta, phi, t, algo)
    ctx.profile_reset(); ctx.profile(True)
    for _ in range(10):
        nlc.ilt_reconstruct(theta, phi, t, algo)
    torch.cuda.synchronize(); ctx.profile(False)
    name = "ilt_fourier_kernel" if algo == "fourier" else "ilt_dehoog_kernel"
    p = ctx.profile_read()[name]
    ms = p["total_ms"] / p["launches"]
    nbytes = N * (2 * d * S + d) * 8
    out["kernels"].append(dict(algo=algo, terms=S, avg_ms=ms, algorithmic_bytes=nbytes, GBps=nbytes / ms / 1e6,
                               frac_hbm_peak=nbytes / ms / 1e6 / 8000.0))
    del theta, phi

import bench
def planner_rate(algo, S, steps):
    torch.manual_seed(0)
    model = nlc.NeuralLaplaceModel(5, 1, 5, hidden_units=128, s_recon_terms=S, ilt_algorithm=algo, state_mean=np.zeros(5),
                                   state_std=np.array([2.88646771, 11.54556671, 0.70729307, 0.70692035, 17.3199048]),
                                   action_mean=np.array([0]), action_std=np.array([1.5]), normalize=True, normalize_time=True).double()
    with torch.no_grad():
        model.laplace_rep_func.linear_tanh_stack[4].bias[5 * S:] += -3.0
    model = model.to("cuda")
    mppi = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost("oderl-cartpole"), 5, nlc.noise_sigma(1), 16384, 40, "cuda",
                         lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0, noise_rng="philox",
                         U_init=torch.zeros(40, 1, dtype=torch.float64))
    st, ab = nlc.initial_state("oderl-cartpole"), torch.zeros(4, 1, dtype=torch.float64)
    with torch.no_grad():
        for _ in range(2):
            mppi.command(st, ab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            mppi.command(st, ab).cpu()
        torch.cuda.synchronize()
    return dict(algo=algo, terms=S, fused=bool(mppi.fused), steps_per_s=steps / (time.perf_counter() - t0))
out["planner"] = [planner_rate("fourier", 17, 20), planner_rate("fourier", 33, 20), planner_rate("dehoog", 33, 5)]
print(json.dumps(out))
