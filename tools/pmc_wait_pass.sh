#!/bin/bash
# One more PMC pass over the headline bench: where do the cycles go that are neither MFMA nor VALU?  (wave wait / issue counters;
# two groups, separate runs, --kernel-trace only)   gpurun -- tools/pmc_wait_pass.sh r3
TAG=${1:-r3}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_WAVES"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/${TAG}_wait_g$i -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ilt --no-other-configs --no-sliced-encoder > $OUT/${TAG}_wait_g$i.log 2>&1 || echo "group $i failed: $(tail -2 $OUT/${TAG}_wait_g$i.log)"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${TAG}_wait_g*/*/*_counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "gru_encode" if "gru_encode" in n else ("nl_rollout" if "nl_rollout" in n else None)
        if k: per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items(): acc[k][c].append(v)
for k, cs in acc.items():
    o = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k, {c: f"{v:.4g}" for c, v in sorted(o.items())})
PY
