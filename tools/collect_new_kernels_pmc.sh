#!/bin/bash
# PMC passes (separate runs per counter group, --kernel-trace only) for the kernels added in round 3:
# ilt_linear_stream (Fourier kernel's LIN instance), ilt_dehoog_bwd, ilt_linear_slot (+ the representation launch beside it:
# the staged planner path of a fixed_tablot model, which hidden width 128 only takes with linear_fused = 0).
#   NLC_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 900 -- "NLC_COMMIT=$NLC_COMMIT tools/collect_new_kernels_pmc.sh r3"
set -o pipefail
TAG=${1:-r3}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
DEV="$(python -c "import torch;print(torch.cuda.get_device_name(0))" 2>/dev/null)"
# provenance as in collect_profiles.sh: the commit the build recorded beside the library, the content hash of the kernel sources
COMMIT="${NLC_COMMIT:-$(python -c "from neurallaplacecontrol_amd import _build_info as b; print(b.COMMIT)" 2>/dev/null || echo unknown)}"
SHA="$(python -c "import __graft_entry__ as g; print(g.csrc_sha())")"
PMC_MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
run() {  # name, counters, program args...
  local name=$1 ctr=$2; shift 2
  timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${TAG}_new_${name} -- python3 "$@" > /dev/null 2>&1 || echo "$name failed"
}
for C in FETCH_SIZE WRITE_SIZE; do
  run lin_$C $C tools/ilt_only.py 655360 fixed_tablot
  run dhb_$C $C tools/dehoog_bwd_bench.py 16384
  CFG5_ALGO=fixed_tablot CFG5_S=17 CFG5_OPTS=linear_fused=0 run slot_$C $C tools/cfg5_breakdown.py
done
run lin_mfma "$PMC_MFMA" tools/ilt_only.py 655360 fixed_tablot
run dhb_mfma "$PMC_MFMA" tools/dehoog_bwd_bench.py 16384
CFG5_ALGO=fixed_tablot CFG5_S=17 CFG5_OPTS=linear_fused=0 run slot_mfma "$PMC_MFMA" tools/cfg5_breakdown.py
# the LIN instances of the rollout kernels (a fixed_tablot model at the headline shape, default path)
for C in FETCH_SIZE WRITE_SIZE; do
  CFG5_ALGO=fixed_tablot CFG5_S=17 run linroll_$C $C tools/cfg5_breakdown.py
done
CFG5_ALGO=fixed_tablot CFG5_S=17 run linroll_mfma "$PMC_MFMA" tools/cfg5_breakdown.py
python tools/pmc_summarize.py --commit "$COMMIT" --csrc_sha "$SHA" --device "$DEV" $OUT/${TAG}_new_linroll_* > $OUT/${TAG}_pmc_lin_rollout.json
python tools/pmc_summarize.py --commit "$COMMIT" --csrc_sha "$SHA" --device "$DEV" $OUT/${TAG}_new_lin_* $OUT/${TAG}_new_dhb_* > $OUT/${TAG}_pmc_new_ilt.json
python tools/pmc_summarize.py --commit "$COMMIT" --csrc_sha "$SHA" --device "$DEV" $OUT/${TAG}_new_slot_* > $OUT/${TAG}_pmc_linear_planner.json
CFG5_ALGO=fixed_tablot CFG5_S=17 CFG5_OPTS=linear_fused=0 timeout -k 10 120 python tools/cfg5_breakdown.py > $OUT/${TAG}_linear_planner_breakdown.txt 2>/dev/null
cat $OUT/${TAG}_linear_planner_breakdown.txt
python - <<PY
import json
for f in ("$OUT/${TAG}_pmc_new_ilt.json", "$OUT/${TAG}_pmc_linear_planner.json"):
    d = json.load(open(f))
    for k, v in d.items():
        if k != "_meta":
            print(f.split("/")[-1], k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("launches", "hbm_bytes_per_launch", "valu_active_frac", "mfma_util")})
PY
