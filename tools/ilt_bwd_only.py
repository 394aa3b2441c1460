"""Times the stand-alone Fourier ILT backward kernel: N points, d=5, S=17 (algorithmic bytes 4dS*8 per point)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
if len(sys.argv) > 2:  # an alternative library build (-DNLC_ILT_EXPERIMENTS=1: NLC_ILT_ROWS / NLC_ILT_DEPTH), before the first ctx exists
    from neurallaplacecontrol_amd import _lib
    _lib.use_library(sys.argv[2])
from neurallaplacecontrol_amd.laplace import default_ctx
N = int(sys.argv[1]) if len(sys.argv) > 1 else 655360
d, S = 5, 17
g = torch.Generator(device="cuda").manual_seed(1)
theta = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi).requires_grad_()
phi = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2) * 0.99).requires_grad_()
t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
gx = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
ctx = default_ctx(0)
for it in range(23):
    if it == 3:
        ctx.profile_reset(); ctx.profile(True)
    x = nlc.ilt_reconstruct(theta, phi, t)
    torch.autograd.grad(x, (theta, phi), gx)
torch.cuda.synchronize()
ctx.profile(False)
p = ctx.profile_read()["ilt_fourier_bwd_kernel"]
ms = p["total_ms"] / p["launches"]
by = N * 4 * d * S * 8
print(f"bwd avg ms {ms:.4f}  points {N}  algorithmic bytes {by}  {by / ms / 1e6:.0f} GB/s")
