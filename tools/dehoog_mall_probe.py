"""is the stand-alone de Hoog row kernel's 1.82x FETCH_SIZE traffic HBM traffic or Infinity-Cache (MALL) hits?
The kernel is timed over the SAME input 20 times at a sweep of N: below ~90 MB the (theta, phi) rows stay resident in the 256 MB
MALL between launches, at 655 360 points (1.76 GB) every launch streams from HBM.  If the time per row does not move across the
residency boundary the kernel is not bound by where its re-reads land.  A bare read stream of the same bytes (torch sum over the
same buffers) is timed beside it as the memory-only yardstick.  python tools/dehoog_mall_probe.py [S]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neurallaplacecontrol_amd as nlc
from neurallaplacecontrol_amd.laplace import default_ctx

S = int(sys.argv[1]) if len(sys.argv) > 1 else 33
d = 5
ctx = default_ctx(0)
rows = []
for N in (8192, 16384, 32768, 65536, 131072, 327680, 655360, 1310720):
    g = torch.Generator(device="cuda").manual_seed(1)
    theta = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * np.pi
    phi = (torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * (np.pi / 2) * 0.9
    t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
    for _ in range(5):
        nlc.ilt_reconstruct(theta, phi, t, "dehoog")
    ctx.profile_reset(); ctx.profile(True)
    for _ in range(20):
        nlc.ilt_reconstruct(theta, phi, t, "dehoog")
    torch.cuda.synchronize()
    ctx.profile(False)
    p = ctx.profile_read()["ilt_dehoog_kernel"]
    ms = p["total_ms"] / p["launches"]
    # memory-only yardstick: one pass over the same two buffers
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        theta.sum(); phi.sum()
    e0.record()
    for _ in range(20):
        theta.sum(); phi.sum()
    e1.record(); torch.cuda.synchronize()
    rd = e0.elapsed_time(e1) / 20
    by = N * (2 * d * S + d) * 8
    rows.append(dict(N=N, S=S, input_MB=round(N * 2 * d * S * 8 / 1e6, 1), kernel_ms=round(ms, 4), ns_per_row=round(ms * 1e6 / (N * d), 3),
                     algorithmic_GBps=round(by / ms / 1e6), read_stream_ms=round(rd, 4)))
    print(rows[-1], file=sys.stderr, flush=True)
    del theta, phi, t
    torch.cuda.empty_cache()
print(json.dumps(dict(what="de Hoog row kernel, time per (point, dim) row vs input residency (MALL 256 MB)", rows=rows)))
