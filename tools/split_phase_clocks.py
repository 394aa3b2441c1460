"""Where a wave of the LATENCY-SPLIT bodies spends its shader clocks.  Four waves share a 16-sample tile and meet
at two / three workgroup barriers per model evaluation; a tools-only build of the library (`make -C neurallaplacecontrol_amd/csrc
variant` -> tools/_libnlc_phase.so) stamps s_memtime at the phase boundaries and sums per wave index, so that what a wave COMPUTES
(layer GEMMs, tanh, sphere epilogue, tail) is separated from what it WAITS for (barriers = the slowest of the four waves, the poll
for latents of another workgroup).  Three workloads, one table each:

  fused   the chain of the one-launch planner body at one 8-GPU shard (K = 2048, T = 40; nl_plan_fused_kernel)
  split   the same rollout as a launch of its own (rollout_variant 2, nl_rollout_split_kernel), no encoder beside it
  rep     the staged de Hoog planner's per-step representation launch (cfg5: K = 16384, S = 33; nl_repfunc_split_kernel)

The MFMA column is 64 clocks per MFMA the phase issues (one wave alone on its SIMD): clocks beyond it are VALU issue, stalls or
waiting.  Stamps cost ~40 clocks each and pin the schedule at the phase boundaries: the totals are a few per cent above the product
build's."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurallaplacecontrol_amd import _lib
LIB = os.environ.get("NLC_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_libnlc_phase.so"))
if not os.path.exists(LIB):
    sys.exit(f"{LIB} is missing: build it with `make -C neurallaplacecontrol_amd/csrc variant` (a tools-only build of the library)")
_lib.use_library(LIB)
import neurallaplacecontrol_amd as nlc
import bench

PHASES = ["head (state / latent operands)", "layer 1 + tanh", "barrier 1", "layer 2 GEMM", "tanh 2 + LDS store", "barrier 2",
          "layer 3 GEMM", "sphere epilogue (+ ILT MFMAs)", "barrier 3", "state update + costs (tail)", "wait for latents (poll)", "other"]
h = ctypes.CDLL(LIB)
out = (ctypes.c_ulonglong * 64)()


def read(name):
    fn = getattr(h, name)
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert fn(out) == 0
    return list(out)


def planner(K, T, algo, S, opts):
    d, nu, A = 5, 1, 3.0
    model = bench.synthetic_state_dict(d, nu, S, algo=algo).to("cuda")
    p = nlc.MPPIDelay(nlc.NLDynamics(model, 0.05), nlc.EnvCost(bench.ENV), d, nlc.noise_sigma(nu), K, T, "cpu", lambda_=1.0,
                      u_min=torch.tensor(-A), u_max=torch.tensor(A), u_scale=A, noise_rng="philox", seed=0,
                      U_init=torch.zeros(T, nu, dtype=torch.float64), store_rollouts=False, planner_options=opts)
    return p


def table(name, reader, K, T, algo, S, opts, mfma_per_wave, n_cmd=30):
    """mfma_per_wave: per phase, MFMAs one wave issues per model evaluation (list of four per-wave lists or one list)."""
    p = planner(K, T, algo, S, opts)
    st, ab = nlc.initial_state(bench.ENV, torch.Generator().manual_seed(0)), torch.zeros(4, 1, dtype=torch.float64)
    with torch.no_grad():
        for _ in range(8):
            p.command(st, ab)
        torch.cuda.synchronize()
        read(reader)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_cmd):
            p.command(st, ab)
        e1.record()
        torch.cuda.synchronize()
        v = read(reader)
    evals = v[52]  # model evaluations (tile-steps) counted by wave 0
    rows = []
    for ph, nm in enumerate(PHASES):
        per_wave = [v[w * 12 + ph] / evals if evals else 0.0 for w in range(4)]
        mf = [mfma_per_wave[w][ph] * 64 for w in range(4)]
        if max(per_wave) < 0.5:
            continue
        rows.append({"phase": nm, "clk_wave0..3": [round(c, 0) for c in per_wave], "mean_clk": round(sum(per_wave) / 4, 0),
                     "mfma_clk_wave0..3": mf})
    tot = [sum(v[w * 12 + ph] for ph in range(12)) / evals if evals else 0.0 for w in range(4)]
    waits = [sum(v[w * 12 + ph] for ph in (2, 5, 8, 10)) / evals if evals else 0.0 for w in range(4)]
    mf_tot = [sum(mfma_per_wave[w]) * 64 for w in range(4)]
    clock_ghz = 2.4
    return {"workload": name, "K": K, "T": T, "ilt": f"{algo} S={S}", "options": opts, "body": p.rollout_body, "commands": n_cmd,
            "ms_per_command_with_stamps": round(e0.elapsed_time(e1) / n_cmd, 4), "tile_evaluations": int(evals),
            "clk_per_evaluation_wave0..3": [round(t, 0) for t in tot], "us_per_evaluation_at_2.4GHz": round(max(tot) / clock_ghz / 1e3, 2),
            "waiting_clk_wave0..3 (barriers + poll)": [round(t, 0) for t in waits], "mfma_clk_wave0..3": mf_tot, "phases": rows}


def mfma_table(HT, NT3, ilt_mfma):
    """MFMAs per wave and phase of one model evaluation: TW = HT / 4 output tiles per wave in layers 1 / 2, layer-3 tiles
    j = wave + 4 i (a wave short of one recomputes the last tile); the epilogue's ILT MFMAs: two per owned tile."""
    KS, TW, NTW = HT * 4, HT // 4, (NT3 + 3) // 4
    res = []
    for w in range(4):
        own = len([i for i in range(NTW) if w + 4 * i < NT3])
        row = [0] * 12
        row[1], row[3], row[6], row[7] = 2 * TW, KS * TW, KS * NTW, (2 * own if ilt_mfma else 0)
        res.append(row)
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["fused", "split", "rep"]
    res = []
    if "fused" in which:
        res.append(table("fused one-launch body, one 8-GPU shard: the chain (rollout role)", "nlc_debug_split_clocks_fused", 2048, 40,
                         "fourier", 17, {"rollout_variant": 3}, mfma_table(8, 11, True)))
    if "split" in which:
        res.append(table("latency-split rollout as its own launch (no encoder beside it)", "nlc_debug_split_clocks_nl", 2048, 40,
                         "fourier", 17, {"rollout_variant": 2}, mfma_table(8, 11, True)))
    if "rep" in which:
        res.append(table("staged de Hoog planner (cfg5): per-step representation launch", "nlc_debug_split_clocks_rep", 16384, 40,
                         "dehoog", 33, {"dehoog_chain": 0, "dehoog_streams": 2}, mfma_table(8, 21, False), n_cmd=10))
        res.append(table("the same on ONE stream (nothing beside the launch)", "nlc_debug_split_clocks_rep", 16384, 40,
                         "dehoog", 33, {"dehoog_chain": 0, "dehoog_streams": 1}, mfma_table(8, 21, False), n_cmd=10))
    print(json.dumps(res, indent=1))
