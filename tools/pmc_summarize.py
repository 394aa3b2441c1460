"""Reduce rocprofv3 --pmc counter_collection.csv files to per-kernel means (JSON on stdout).

    python tools/pmc_summarize.py gpurun_out/pmc_h_FETCH_SIZE gpurun_out/pmc_h_WRITE_SIZE gpurun_out/pmc_h_mfma

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB; on gfx950 FETCH_SIZE counts 32 B per 64 B request and is
doubled here (calibrated with tools/pmc_calib.hip, profiles/r1_pmc_calib_*.csv; MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import glob
import json
import sys

KEYS = ("gru_encode", "nl_rollout", "ilt_fourier_bwd", "ilt_fourier", "ilt_dehoog", "perturb", "weight_partial",
        "oracle_rollout", "rnn_encode", "rnn_rollout", "merge")


def short(name):
    for k in KEYS:
        if k in name:
            return k
    return None


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
            per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
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
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
