"""Why is nl_rollout_kernel slower behind the int8-sliced encoder than behind the FP64 one (VERDICT r5 weak 2 / item 1a:
1.10 -> 1.15-1.26 ms on an untouched kernel)?  The headline planner (cartpole, K = 16384, H = 40) on the phase-clock build of the
library (`make -C neurallaplacecontrol_amd/csrc variant` -> tools/_libnlc_phase.so): per case the rollout launch's hipEvent duration
beside the shader clocks its wavefronts counted (s_memtime) -- clocks / duration is the clock the kernel ran at; more CLOCKS means
the kernel waited (cache contents), the same clocks in more TIME means the chip ran slower (power management).  Cases: either
encoder in front; 1 ms of idle GPU between the launches ("dbg_gap_us"); a 64 MB read sweep between them ("dbg_l2_mb": the L2s
hold 32 MB); the cases take turns, three rounds."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurallaplacecontrol_amd import _lib
LIB = os.environ.get("NLC_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libnlc_phase.so"))
if not os.path.exists(LIB):
    sys.exit(f"{LIB} is missing: build it with `make -C neurallaplacecontrol_amd/csrc variant`")
_lib.use_library(LIB)
import neurallaplacecontrol_amd as nlc
import bench

env, d, nu, A, K, T, S = "oderl-cartpole", 5, 1, 3.0, 16384, 40, 17
model = bench.synthetic_state_dict(d, nu, S).to("cuda")
h = ctypes.CDLL(LIB)
h.nlc_debug_phase_clocks.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
out = (ctypes.c_ulonglong * 16)()
CASES = [("fp64 encoder -> rollout", {}), ("i8 encoder -> rollout", {"gru_gemm": 1}),
         ("fp64 encoder -> 1 ms idle -> rollout", {"dbg_gap_us": 1000}), ("i8 encoder -> 1 ms idle -> rollout", {"gru_gemm": 1, "dbg_gap_us": 1000}),
         ("fp64 encoder -> 64 MB read sweep -> rollout", {"dbg_l2_mb": 64}), ("i8 encoder -> 64 MB read sweep -> rollout", {"gru_gemm": 1, "dbg_l2_mb": 64}),
         ("i8 encoder -> 100 us idle -> rollout", {"gru_gemm": 1, "dbg_gap_us": 100})]
planners = []
for name, opts in CASES:
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(env), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                      u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                      U_init=torch.zeros(T, nu, dtype=torch.float64), store_rollouts=True, planner_options=dict(opts, rollout_variant=1))
    planners.append(p)
st, ab = nlc.initial_state(env, torch.Generator().manual_seed(0)), torch.zeros(4, nu, dtype=torch.float64)
rows = [dict(case=n, options=o, rollout_ms=[], encoder_ms=[], clk_per_wave=[], mhz=[]) for n, o in CASES]
with torch.no_grad():
    for p in planners:
        for _ in range(40):
            p.command(st, ab)
    torch.cuda.synchronize()
    for rnd in range(3):
        for p, row in zip(planners, rows):
            for _ in range(5):
                p.command(st, ab)
            torch.cuda.synchronize()
            assert h.nlc_debug_phase_clocks(out) == 0  # clears the sums
            p.ctx.profile_reset()
            p.ctx.profile(True)
            n_cmd = 30
            for _ in range(n_cmd):
                p.command(st, ab)
            torch.cuda.synchronize()
            p.ctx.profile(False)
            prof = p.ctx.profile_read()
            assert h.nlc_debug_phase_clocks(out) == 0
            waves = out[10]
            clk = sum(out[i] for i in range(10)) / waves if waves else 0.0  # clocks per wavefront and launch (its T steps)
            r_ms = prof["nl_rollout_kernel"]["total_ms"] / prof["nl_rollout_kernel"]["launches"]
            e_ms = prof["gru_encode_kernel"]["total_ms"] / prof["gru_encode_kernel"]["launches"]
            row["rollout_ms"].append(round(r_ms, 4))
            row["encoder_ms"].append(round(e_ms, 4))
            row["clk_per_wave"].append(round(clk))
            row["mhz"].append(round(clk / (r_ms * 1e3)))
for row in rows:
    for k in ("rollout_ms", "encoder_ms", "clk_per_wave", "mhz"):
        row[k + "_median"] = float(np.median(row[k]))
print(json.dumps(dict(workload=f"cartpole fourier S={S} K={K} H={T}, wave-per-tile rollout, phase-clock build (stamps cost ~1 %)",
                      note="mhz = s_memtime clocks a wavefront counted over its T steps / the launch's hipEvent duration", cases=rows), indent=1))
