"""Runs only the stand-alone ILT stream kernel (for rocprofv3 --pmc passes): N points, d=5, S=17 (16 for stehfest).
python tools/ilt_only.py [N] [fourier|fixed_tablot|stehfest] [S] [library.so]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
if len(sys.argv) > 4:  # an alternative library build (e.g. -DNLC_ILT_EXPERIMENTS=1), before the first ctx exists
    from neurallaplacecontrol_amd import _lib
    _lib.use_library(sys.argv[4])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 655360
algo = sys.argv[2] if len(sys.argv) > 2 else "fourier"
d, S = 5, (int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else (16 if algo == "stehfest" else 17))
g = torch.Generator(device="cuda").manual_seed(1)
theta = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
phi = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2)
t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
from neurallaplacecontrol_amd.laplace import default_ctx
ctx = default_ctx(0)
for _ in range(3):
    x = nlc.ilt_reconstruct(theta, phi, t, algo)
ctx.profile_reset(); ctx.profile(True)
for _ in range(20):
    x = nlc.ilt_reconstruct(theta, phi, t, algo)
torch.cuda.synchronize()
ctx.profile(False)
prof = ctx.profile_read()
p = prof["ilt_fourier_kernel"] if algo == "fourier" else (prof.get("ilt_linear_kernel") or prof["ilt_linear_stream_kernel"])
ms = p["total_ms"] / p["launches"]
nbytes = N * (2 * d * S + d) * 8
print(algo, "avg ms", ms, "dbg", os.environ.get("NLC_ILT_DBG", "0"))
print("points", N, "algorithmic bytes per launch", nbytes, "=", round(nbytes / ms / 1e6), "GB/s")
