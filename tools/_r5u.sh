#!/bin/bash
# instruction-cache counters of the two encoders, tools only
OUT=gpurun_out/r5u
mkdir -p $OUT
export TMPDIR=/tmp
G1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES"
G2="SQ_WAIT_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_WAVE_CYCLES"
i=0
for G in "$G1" "$G2"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -- python3 tools/i8_gemm_probe.py --steps 4 --windows 70000 > $OUT/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $OUT/g$i.log)"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/*/*_counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "gru_i8" if "gru_encode_i8" in n else ("gru_f64" if "gru_encode_kernel" in n else ("rollout" if "nl_rollout_kernel" in n else None))
        if k and int(r.get("Grid_Size", "0") or 0) >= 65536: per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items(): acc[k][c].append(v)
for k, cs in sorted(acc.items()):
    o = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k, {c: f"{v:.4g}" for c, v in sorted(o.items())})
PY
