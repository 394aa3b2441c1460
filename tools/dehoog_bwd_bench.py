"""de Hoog ILT backward (ilt_dehoog_bwd_kernel, reverse mode through the QD table with an HBM tape) timed beside the
same recurrences as PyTorch-ROCm tensor ops under autograd (what the package did before round 3), S = 33, d = 5:
a training-size batch and the bench's N = K*T.  One JSON line on stdout."""
import json, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurallaplacecontrol_amd import _lib
if os.environ.get("NLC_LIB_PATH"):  # tools only: another build of the same library (A/B of kernel variants)
    _lib.use_library(os.environ["NLC_LIB_PATH"])
import neurallaplacecontrol_amd as nlc


def torch_ops(theta, phi, t, desc):
    S = theta.shape[-1]
    M = (S - 1) // 2
    t = t.view(-1, 1)
    T = desc.scale * t
    gamma = desc.alpha - math.log(desc.tol) / (desc.scale * T)
    r = torch.tan(phi / 2.0 + math.pi / 4.0)
    fp = torch.complex(r * torch.cos(theta), r * torch.sin(theta))
    a = [fp[..., 0] / 2.0] + [fp[..., i] for i in range(1, S)]
    q = [a[i + 1] / a[i] for i in range(2 * M)]
    e = [torch.zeros_like(a[0]) for _ in range(S)]
    d = [a[0], -q[0]]
    for rr in range(1, M + 1):
        mr = 2 * (M - rr) + 1
        e = [q[i + 1] - q[i] + e[i + 1] for i in range(mr)]
        d.append(-e[0])
        if rr != M:
            q = [q[i + 1] * e[i + 1] / e[i] for i in range(mr - 1)]
            d.append(-q[0])
    ang = math.pi * (t / T)
    z = torch.complex(torch.cos(ang), torch.sin(ang))
    A_prev, A_cur = torch.zeros_like(d[0]), d[0]
    B_prev, B_cur = torch.ones_like(d[0]), torch.ones_like(d[0])
    for i in range(1, 2 * M):
        A_prev, A_cur = A_cur, A_cur + d[i] * A_prev * z
        B_prev, B_cur = B_cur, B_cur + d[i] * B_prev * z
    brem = (1.0 + (d[2 * M - 1] - d[2 * M]) * z) / 2.0
    rem = brem * (torch.sqrt(1.0 + d[2 * M] * z / brem) - 1.0)
    res = (A_cur + rem * A_prev) / (B_cur + rem * B_prev)
    return torch.exp(gamma * t) / T * res.real


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = []
S, d = 33, 5
desc = _lib.ilt_desc("dehoog", S)
SIZES = [(int(a), False) for a in sys.argv[1:]] or [(1024, True), (16384, True), (655360, False)]  # argv: sizes, HIP only
for N, with_torch in SIZES:
    g = torch.Generator(device="cuda").manual_seed(N)
    theta = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 3.0).requires_grad_()
    phi = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 1.2).requires_grad_()
    t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
    gx = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
    row = dict(N=N, d=d, S=S, rows=N * d)
    row["hip_forward_backward_ms"] = timed(lambda: torch.autograd.grad(nlc.ilt_reconstruct(theta, phi, t, "dehoog"), (theta, phi), gx), 5)
    with torch.no_grad():
        row["hip_forward_only_ms"] = timed(lambda: nlc.ilt_reconstruct(theta, phi, t, "dehoog"), 5)
    if with_torch:
        row["torch_ops_forward_backward_ms"] = timed(lambda: torch.autograd.grad(torch_ops(theta, phi, t, desc), (theta, phi), gx), 2)
        ga = torch.autograd.grad(nlc.ilt_reconstruct(theta, phi, t, "dehoog"), (theta, phi), gx)
        gb = torch.autograd.grad(torch_ops(theta, phi, t, desc), (theta, phi), gx)
        row["median_rel_diff_vs_torch_ops"] = float(((ga[0] - gb[0]).abs() / (gb[0].abs() + 1e-300)).median())
    out.append(row)
    print(row, file=sys.stderr, flush=True)
print(json.dumps(dict(metric="de Hoog ILT backward, one MI355X, f64", results=out)))
