"""Command time of the three planner paths at BASELINE configs[1] size: fused EnvCost, fused dynamics + a running_cost
callable (the harness's state_constraint branch), and the generic path (dynamics as an opaque closure)."""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import neurallaplacecontrol_amd as nlc
import bench
from oracle import envs as oenvs
model = bench.synthetic_state_dict(5, 1, 17).cuda()
cost = oenvs.cartpole_cost_variant(state_constraint=True)
kw = dict(lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0, noise_rng="philox")
st, ab = nlc.initial_state("oderl-cartpole"), torch.zeros(4, 1, dtype=torch.float64)
dyn = nlc.NLDynamics(model, 0.05)
for name, d, c in (("fused EnvCost", dyn, nlc.EnvCost("oderl-cartpole")), ("fused dynamics + cost callable", dyn, cost),
                   ("generic (closure dynamics)", (lambda s, w: dyn(s, w)), cost)):
    p = nlc.MPPIDelay(d, c, 5, nlc.noise_sigma(1), 16384, 40, "cuda", **kw)
    with torch.no_grad():
        for _ in range(3): p.command(st, ab).cpu()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): p.command(st, ab).cpu()
        torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms / command (K=16384, T=40)")
