#!/bin/bash
# Wave wait / issue counters of the two encoders (FP64 MFMA vs int8-sliced, gru_gemm = 1): four PMC groups, each in its own run
# (--kernel-trace only), over tools/i8_gemm_probe.py; per-kernel means on stdout (profiles/r5_i8_gemm.md, section 3).
#   gpurun -- "bash tools/i8_pmc.sh"
OUT=gpurun_out/i8_pmc
mkdir -p $OUT
export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G2="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES"
G4="SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
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
        k = "gru_i8" if "gru_encode_i8" in n else ("gru_f64" if "gru_encode_kernel" in n else None)
        if k and int(r.get("Grid_Size", "0") or 0) >= 655360 * 4: per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        for c, v in cs.items(): acc[k][c].append(v)
for k, cs in sorted(acc.items()):
    o = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k, {c: f"{v:.4g}" for c, v in sorted(o.items())})
PY
