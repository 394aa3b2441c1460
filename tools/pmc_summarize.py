"""Reduce rocprofv3 --pmc counter_collection.csv files to per-kernel means (JSON on stdout).

    python tools/pmc_summarize.py [--commit HASH] [--device NAME] gpurun_out/pmc_h_FETCH_SIZE gpurun_out/pmc_h_WRITE_SIZE ...

The "_meta" entry records where the numbers come from (commit, device, date) and, per key, the rocprof kernel names that
were folded into it: bench.py refuses a summary whose kernel names are not the library's (a stale file).

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; on gfx950 FETCH_SIZE counts 32 B per 64 B request and is
doubled here (calibrated with tools/pmc_calib.hip, profiles/r1_pmc_calib_*.csv; MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import glob
import json
import sys

KEYS = ("gru_encode", "nl_plan_fused", "nl_rollout", "nl_repfunc", "nl_dehoog_chain", "ilt_fourier_bwd", "ilt_fourier", "ilt_dehoog_bwd", "ilt_dehoog",
        "ilt_linear_slot", "ilt_linear_bwd", "ilt_linear", "perturb",
        "weight_tile", "weight_rank", "weight_chunk", "weight_final", "oracle_rollout", "rnn_encode", "rnn_rollout", "merge", "step_tail")


def short(name):
    if "ilt_fourier_kernel" in name and "true>" in name:
        return "ilt_linear_stream"  # the Fourier kernel's LIN instance (fixed Talbot / Stehfest)
    if "ilt_fourier_bwd_rows_kernel" in name:
        return "ilt_fourier_bwd"  # round 6: the row-per-lane kernels (direct global -> LDS tile loads)
    if "ilt_fourier_rows_kernel" in name:
        return "ilt_linear_stream" if ", true," in name else "ilt_fourier"
    for k in KEYS:
        if k in name:
            return k
    return None


def main():
    import datetime

    argv, meta = sys.argv[1:], {"date": datetime.date.today().isoformat()}
    while argv and argv[0].startswith("--"):
        meta[argv[0][2:]] = argv[1]
        argv = argv[2:]
    names = collections.defaultdict(set)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in argv:
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    names[k].add(r["Kernel_Name"].split("(")[0][:120])
                    per_dispatch[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            for (k, _), cs in per_dispatch.items():
                for c, v in cs.items():
                    acc[k][c].append(v)
    out = {}
    for k, cs in acc.items():
        o = {c: sum(v) / len(v) for c, v in cs.items()}
        o["launches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
            o["hbm_bytes_per_launch"] = (2.0 * o["FETCH_SIZE"] + o["WRITE_SIZE"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "GRBM_GUI_ACTIVE" in o:
            cycles = o["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            o["mfma_util"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024)
            o["valu_active_frac"] = 4.0 * o["SQ_ACTIVE_INST_VALU"] / (cycles * 1024)
            if o.get("SQ_INSTS_VALU_MFMA_F64"):
                o["cycles_per_mfma"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["SQ_INSTS_VALU_MFMA_F64"]
        out[k] = o
    meta["kernel_names"] = {k: sorted(v) for k, v in names.items()}
    out["_meta"] = meta
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
