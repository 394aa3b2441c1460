import sys, torch
sys.path.insert(0, '.')
import neurallaplacecontrol_amd as nlc
N, d, S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 5, 33
g = torch.Generator(device="cuda").manual_seed(1)
th = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 3.0).requires_grad_()
ph = ((torch.rand(N, d, S, dtype=torch.float64, device="cuda", generator=g) * 2 - 1) * 1.2).requires_grad_()
t = torch.full((N,), 0.125, dtype=torch.float64, device="cuda")
gx = torch.randn(N, d, dtype=torch.float64, device="cuda", generator=g)
ga = torch.autograd.grad(nlc.ilt_reconstruct(th, ph, t, "dehoog"), (th, ph), gx)
torch.cuda.synchronize()
print("ok", float(ga[0].abs().sum()))
