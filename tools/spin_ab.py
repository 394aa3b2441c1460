"""A/B of the host hand-over of the action (tools only): host_spin 0 = hipStreamSynchronize, 1 = spin on the pinned sequence
word the merge kernel stores.  python tools/spin_ab.py"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc

d, nu = 5, 1
model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0))
for rep in range(2):
    for K, T in ((1024, 20), (2048, 40), (16384, 40)):
        for spin in (0, 1):
            p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                              device="cpu", compute_device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0),
                              u_scale=3.0, noise_rng="philox", seed=0, U_init=torch.zeros(T, nu, dtype=torch.float64),
                              planner_options={"host_spin": spin})
            ab = torch.zeros(4, nu, dtype=torch.float64)
            n = 300 if K <= 2048 else 60
            for i in range(20 + n):
                if i == 20:
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                a = p.command(state, ab)
                ab = torch.roll(ab, -1, 0); ab[-1] = a
            torch.cuda.synchronize()
            print(json.dumps(dict(K=K, T=T, host_spin=spin, ms_per_step=round((time.perf_counter() - t0) / n * 1e3, 4))), flush=True)
