"""A few commands of the two-launch (2) and fused (3) rollout bodies at K = 2048 for rocprofv3 --pmc runs (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d, nu = 5, 1
model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0))
for variant in (2, 3):
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K,
                      horizon=bench.HORIZON, device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0),
                      u_scale=3.0, noise_rng="philox", seed=0, U_init=torch.zeros(bench.HORIZON, nu, dtype=torch.float64),
                      planner_options={"rollout_variant": variant})
    ab = torch.zeros(4, nu, dtype=torch.float64)
    for _ in range(6):
        a = p.command(state, ab)
        ab = torch.roll(ab, -1, 0)
        ab[-1] = a.cpu()
    torch.cuda.synchronize()
