#!/bin/bash
# The clock each planner kernel runs at: GRBM_GUI_ACTIVE (GPU-active cycles, summed over the 8 XCDs) of every dispatch over its
# duration from the same rocprofv3 record -- the headline planner with either encoder (tools/i8_gemm_probe.py).
#   gpurun -- "bash tools/kernel_clocks.sh"
OUT=gpurun_out/kernel_clocks
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/run -- python3 tools/i8_gemm_probe.py --steps 12 --windows 70000 > $OUT/run.log 2>&1 || { echo "failed: $(tail -3 $OUT/run.log)"; exit 1; }
python3 - <<PY
import csv, glob, collections, statistics
f = glob.glob("$OUT/run/*/*_counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
cols = rows[0].keys()
print("columns:", [c for c in cols if "imestamp" in c or c in ("Kernel_Name", "Counter_Name", "Counter_Value", "Dispatch_Id")])
per = collections.defaultdict(lambda: dict(cycles=0.0))
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    d = per[r["Dispatch_Id"]]
    d["cycles"] += float(r["Counter_Value"])
    d["name"] = r["Kernel_Name"].split("(")[0][:60]
    if "Start_Timestamp" in r:
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
# durations from the kernel trace when the counter file has none
if not any("ns" in d for d in per.values()):
    t = glob.glob("$OUT/run/*/*_kernel_trace.csv")[0]
    for r in csv.DictReader(open(t)):
        if r["Dispatch_Id"] in per:
            per[r["Dispatch_Id"]]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
order = sorted(per, key=lambda k: int(k))
prev_enc = None
acc = collections.defaultdict(list)
for k in order:
    d = per[k]
    if "ns" not in d or d["ns"] <= 0:
        continue
    mhz = d["cycles"] / 8.0 / d["ns"] * 1e3
    n = d["name"]
    if "gru_encode_i8" in n:
        prev_enc = "i8"; acc["gru_encode_i8_kernel"].append((mhz, d["ns"]))
    elif "gru_encode_kernel" in n and d["ns"] > 1e6:
        prev_enc = "f64"; acc["gru_encode_kernel (FP64)"].append((mhz, d["ns"]))
    elif "nl_rollout_kernel" in n and d["ns"] > 5e5:
        acc[f"nl_rollout_kernel behind the {prev_enc} encoder"].append((mhz, d["ns"]))
for n, v in acc.items():
    print(f"{n:48s} launches {len(v):3d}  median clock {statistics.median(x[0] for x in v):7.0f} MHz  median duration {statistics.median(x[1] for x in v)/1e6:.4f} ms")
PY
