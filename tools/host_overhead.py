"""Host-side cost of one command() (tools only): a tiny population (K = 16, T = 2) makes the GPU work negligible, so the
per-command wall time is the Python / ctypes / launch / synchronisation overhead every command pays."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc

d, nu = 5, 1
model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0))
for K, T in ((16, 2), (1024, 20), (2048, 40)):
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K, horizon=T,
                      device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0), u_scale=3.0,
                      noise_rng="philox", seed=0, U_init=torch.zeros(T, nu, dtype=torch.float64))
    ab = torch.zeros(4, nu, dtype=torch.float64)
    for _ in range(20):
        p.command(state, ab).cpu()
    n = 300
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        a = p.command(state, ab)
        ab = torch.roll(ab, -1, 0); ab[-1] = a.cpu()
    torch.cuda.synchronize(); sync_ms = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        a = p.command(state, ab)
    t_issue = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize(); async_ms = (time.perf_counter() - t0) / n * 1e3
    p.ctx.profile_reset(); p.ctx.profile(True)
    for _ in range(50):
        p.command(state, ab).cpu()
    torch.cuda.synchronize(); p.ctx.profile(False)
    ksum = sum(v["total_ms"] / 50 for v in p.ctx.profile_read().values())
    print(json.dumps(dict(K=K, T=T, ms_per_command_synced=round(sync_ms, 4), ms_host_issue_only=round(t_issue, 4),
                          ms_per_command_unsynced=round(async_ms, 4), kernel_sum_ms=round(ksum, 4))), flush=True)
