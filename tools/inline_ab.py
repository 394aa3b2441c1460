"""A/B of the fused body's in-launch phases on one box (tools only): python tools/inline_ab.py
fused_inline bit mask: 0 separate perturb / weight launches, 1 weights inside, 2 sampling inside, 3 both; device 'cpu' =
action through pinned memory, 'cuda' = action tensor on the device (+ .cpu() by the caller)."""
import sys, os, time, json, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc

K = int(os.environ.get("AB_K", "2048"))
d, nu = 5, 1
model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0))


def make(mask, device):
    return nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K,
                         horizon=bench.HORIZON, device=device, lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0),
                         u_scale=3.0, noise_rng="philox", seed=0, U_init=torch.zeros(bench.HORIZON, nu, dtype=torch.float64),
                         planner_options={"rollout_variant": 3, "fused_inline": mask}, compute_device="cuda:0")


def run(p, n):
    ab = torch.zeros(4, nu, dtype=torch.float64)
    for _ in range(10):
        a = p.command(state, ab)
        ab = torch.roll(ab, -1, 0); ab[-1] = a.cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        a = p.command(state, ab)
        ab = torch.roll(ab, -1, 0); ab[-1] = a.cpu()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    for device in ("cuda:0", "cpu"):
        for mask in (0, 1, 2, 3):
            p = make(mask, device)
            ms = run(p, 300)
            p.ctx.profile_reset(); p.ctx.profile(True)
            run(p, 40)
            p.ctx.profile(False)
            ks = {k: round(v["total_ms"] / v["launches"], 4) for k, v in p.ctx.profile_read().items()}
            print(json.dumps(dict(K=K, device=device, fused_inline=mask, ms_per_step=round(ms, 4), kernels=ks)), flush=True)
            del p
# where the host time goes (device 'cpu', everything inside the launch)
p = make(3, "cpu")
run(p, 50)
pr = cProfile.Profile()
ab = torch.zeros(4, nu, dtype=torch.float64)
pr.enable()
for _ in range(500):
    a = p.command(state, ab)
    ab = torch.roll(ab, -1, 0); ab[-1] = a.cpu()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue())
