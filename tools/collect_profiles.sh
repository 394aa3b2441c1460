#!/bin/bash
# Collects the per-round profile set on the GPU box into gpurun_out/<tag>_*: bench line, rocprofv3 kernel stats,
# PMC passes (HBM traffic, MFMA / VALU utilisation; each counter group in its own run, --kernel-trace only).
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'tools/collect_profiles.sh r1j'
set -o pipefail
TAG=${1:-r1x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err || exit 1
echo "bench line done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_stdout.log 2>&1 || exit 1
echo "kernel stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  echo "pmc $C done"
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || echo "mfma counter pass failed (continuing)"
python tools/pmc_summarize.py --commit "$(git rev-parse --short HEAD 2>/dev/null || echo "${NLC_COMMIT:-unknown}")" --device "$(python -c "import torch;print(torch.cuda.get_device_name(0))" 2>/dev/null)" $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_mfma > $OUT/${TAG}_pmc_kernels.json
cp $(ls $OUT/${TAG}_stats/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
cut -c1-600 $OUT/${TAG}_bench_line.json
head -12 $OUT/${TAG}_bench_kernel_stats.csv
