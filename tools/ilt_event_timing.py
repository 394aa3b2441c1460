import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import neurallaplacecontrol_amd as nlc
from neurallaplacecontrol_amd.laplace import default_ctx
N, d, S = 655360, 5, 17
g = torch.Generator(device="cuda").manual_seed(1)
theta = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
phi = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2)
t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
ctx = default_ctx(0)
for _ in range(5):
    nlc.ilt_reconstruct(theta, phi, t)
torch.cuda.synchronize()
# (a) per-launch event pairs (library profile)
ctx.profile_reset(); ctx.profile(True)
t0 = time.perf_counter()
for _ in range(20):
    nlc.ilt_reconstruct(theta, phi, t)
t_host = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
ctx.profile(False)
p = ctx.profile_read()["ilt_fourier_kernel"]
print("per-launch pairs avg ms", p["total_ms"] / p["launches"], "host enqueue ms/call", t_host)
# (b) one pair around 20 queued launches, profiling off
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        nlc.ilt_reconstruct(theta, phi, t)
    e1.record()
    torch.cuda.synchronize()
    print("one pair / 20 launches ms", e0.elapsed_time(e1) / 20)
