"""Times the stand-alone de Hoog ILT kernel: N points, d=5, S in {33, 17} (NLC_ILT_DBG=1 selects the alternative
occupancy build of the kernel)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
from neurallaplacecontrol_amd.laplace import default_ctx
N = int(sys.argv[1]) if len(sys.argv) > 1 else 655360
d = 5
ctx = default_ctx(0)
for S in (33, 17):
    g = torch.Generator(device="cuda").manual_seed(1)
    theta = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2) * 0.9
    t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
    for _ in range(3):
        x = nlc.ilt_reconstruct(theta, phi, t, "dehoog")
    ctx.profile_reset(); ctx.profile(True)
    for _ in range(10):
        x = nlc.ilt_reconstruct(theta, phi, t, "dehoog")
    torch.cuda.synchronize()
    ctx.profile(False)
    p = ctx.profile_read()["ilt_dehoog_kernel"]
    ms = p["total_ms"] / p["launches"]
    by = N * (2 * d * S + d) * 8
    print(f"S={S} avg ms {ms:.4f}  {by / ms / 1e6:.0f} GB/s  dbg={os.environ.get('NLC_ILT_DBG', '0')}")
