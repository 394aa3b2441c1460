"""Times one shard's command() with the three rollout bodies (tools only): python tools/fused_probe.py [K ...]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc

Ks = [int(a) for a in sys.argv[1:]] or [2048]
d, nu = 5, 1
model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0))
out = []
for K in Ks:
    for variant, cap in [(1, 0), (2, 0), (3, 0), (3, 64)]:
        if variant == 3 and cap and cap > (K + 15) // 16:
            continue
        opts = {"rollout_variant": variant}
        if cap:
            opts["fused_roll_cap"] = cap
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K,
                          horizon=bench.HORIZON, device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0),
                          u_scale=3.0, noise_rng="philox", seed=0, U_init=torch.zeros(bench.HORIZON, nu, dtype=torch.float64),
                          planner_options=opts, store_rollouts=True)
        ab = torch.zeros(4, nu, dtype=torch.float64)
        def step(ab):
            a = p.command(state, ab)
            ab = torch.roll(ab, -1, 0); ab[-1] = a.cpu(); return ab
        for _ in range(5): ab = step(ab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 100
        for _ in range(n): ab = step(ab)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        p.ctx.profile_reset(); p.ctx.profile(True)
        for _ in range(20): ab = step(ab)
        torch.cuda.synchronize(); p.ctx.profile(False)
        prof = {k: round(v["total_ms"] / v["launches"], 4) for k, v in p.ctx.profile_read().items()}
        rec = dict(K=K, variant=variant, roll_cap=cap, ms_per_command=round(ms, 4), kernels_ms=prof)
        print(json.dumps(rec), flush=True)
        out.append(rec)
