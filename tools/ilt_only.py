"""Runs only the stand-alone ILT Fourier kernel (for rocprofv3 --pmc passes): N points, d=5, S=17."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 655360
d, S = 5, 17
g = torch.Generator(device="cuda").manual_seed(1)
theta = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
phi = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2)
t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
from neurallaplacecontrol_amd.laplace import default_ctx
ctx = default_ctx(0)
for _ in range(3):
    x = nlc.ilt_reconstruct(theta, phi, t)
ctx.profile_reset(); ctx.profile(True)
for _ in range(20):
    x = nlc.ilt_reconstruct(theta, phi, t)
torch.cuda.synchronize()
ctx.profile(False)
p = ctx.profile_read()["ilt_fourier_kernel"]
print("avg ms", p["total_ms"] / p["launches"], "dbg", os.environ.get("NLC_ILT_DBG", "0"))
print("points", N, "algorithmic bytes per launch", N * (2 * d * S + d) * 8)
