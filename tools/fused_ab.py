"""A/B of the fused planner body on one box (tools only): python tools/fused_ab.py [<chain_first_tiles>x<partner_tiles>|none ...]
NLC_LIB_PATH=<other libnlc_hip.so> selects the library; `none` leaves the schedule at the library's auto choice."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import neurallaplacecontrol_amd as nlc
if os.environ.get("NLC_LIB_PATH"):  # tools only: another build of the same library
    from neurallaplacecontrol_amd import _lib as _nlc_lib
    _nlc_lib.use_library(os.environ["NLC_LIB_PATH"])

K = int(os.environ.get("AB_K", "2048"))
d, nu = 5, 1
model = bench.synthetic_state_dict(d, nu, bench.S_TERMS).to("cuda:0")
state = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0))
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for own in sys.argv[1:] or ["none"]:
        opts = {"rollout_variant": 3}
        if os.environ.get("AB_CAP"):
            opts["fused_roll_cap"] = int(os.environ["AB_CAP"])
        if os.environ.get("AB_RATIO"):
            opts["fused_tile_step_ratio"] = float(os.environ["AB_RATIO"])
        if os.environ.get("AB_BPC"):
            opts["fused_blocks_per_cu"] = int(os.environ["AB_BPC"])
        if own != "none":
            opts["fused_chain_first_tiles"] = int(own.split("x")[0])
            opts["fused_partner_tiles"] = int(own.split("x")[1])
        p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), num_samples=K,
                          horizon=bench.HORIZON, device="cuda:0", lambda_=1.0, u_min=torch.tensor(-3.0), u_max=torch.tensor(3.0),
                          u_scale=3.0, noise_rng="philox", seed=0, U_init=torch.zeros(bench.HORIZON, nu, dtype=torch.float64),
                          planner_options=opts)
        ab = torch.zeros(4, nu, dtype=torch.float64)

        def step(ab):
            a = p.command(state, ab)
            ab = torch.roll(ab, -1, 0)
            ab[-1] = a.cpu()
            return ab

        for _ in range(10):
            ab = step(ab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            ab = step(ab)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        p.ctx.profile_reset()
        p.ctx.profile(True)
        for _ in range(50):
            ab = step(ab)
        torch.cuda.synchronize()
        p.ctx.profile(False)
        prof = p.ctx.profile_read()
        k = prof["nl_plan_fused_kernel"]
        print(json.dumps(dict(ratio=os.environ.get("AB_RATIO"), bpc=os.environ.get("AB_BPC"), K=K, own=own, ms_per_command=round(ms, 4),
                              fused_ms=round(k["total_ms"] / k["launches"], 4))), flush=True)
