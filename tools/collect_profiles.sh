#!/bin/bash
# Collects the per-round profile set on the GPU box into gpurun_out/<tag>_*: bench line, rocprofv3 kernel stats,
# PMC passes (HBM traffic, MFMA / VALU utilisation; each counter group in its own run, --kernel-trace only), for the
# headline bench, the cfg5 (de Hoog) planner and one 8-GPU shard (K = 2048, fused body).
#   NLC_COMMIT=$(git rev-parse --short HEAD) /usr/local/graft/bin/gpurun --timeout 1100 -- "NLC_COMMIT=$NLC_COMMIT tools/collect_profiles.sh r2"
set -o pipefail
TAG=${1:-r3x}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# provenance of every summary: the commit __graft_entry__.build() recorded beside the library (the box has no .git) and the
# content hash of the kernel sources (bench.py quotes `roofline.traffic` only from a summary whose hash is the library's)
COMMIT="${NLC_COMMIT:-$(python -c "from neurallaplacecontrol_amd import _build_info as b; print(b.COMMIT)" 2>/dev/null || echo unknown)}"
SHA="$(python -c "import __graft_entry__ as g; print(g.csrc_sha())")"
DEV="$(python -c "import torch;print(torch.cuda.get_device_name(0))" 2>/dev/null)"
timeout -k 10 300 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err || exit 1
echo "bench line done"
timeout -k 10 120 python bench.py --samples 2048 --no-cpu-baseline --no-ilt > $OUT/${TAG}_bench_line_K2048_experiment.json 2>> $OUT/${TAG}_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-sliced-encoder > $OUT/${TAG}_bench_stdout.log 2>&1 || exit 1
echo "kernel stats done"
PMC_MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-sliced-encoder > /dev/null 2>&1 || exit 1
  echo "pmc $C done"
done
timeout -k 10 300 rocprofv3 --pmc $PMC_MFMA --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_mfma -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-sliced-encoder > /dev/null 2>&1 || echo "mfma counter pass failed (continuing)"
python tools/pmc_summarize.py --commit "$COMMIT" --csrc_sha "$SHA" --device "$DEV" $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_mfma > $OUT/${TAG}_pmc_kernels.json
cp $(ls $OUT/${TAG}_stats/*/*_kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
# cfg5 (de Hoog planner, staged path) and the fused small-shard body: their own PMC summaries
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_cfg5_pmc_$C -- python3 tools/cfg5_breakdown.py > /dev/null 2>&1 || echo "cfg5 pmc $C failed"
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_k2048_pmc_$C -- python3 tools/fused_pmc.py > /dev/null 2>&1 || echo "k2048 pmc $C failed"
done
timeout -k 10 200 rocprofv3 --pmc $PMC_MFMA --kernel-trace --output-format csv -d $OUT/${TAG}_cfg5_pmc_mfma -- python3 tools/cfg5_breakdown.py > /dev/null 2>&1 || echo "cfg5 mfma pass failed"
timeout -k 10 200 rocprofv3 --pmc $PMC_MFMA --kernel-trace --output-format csv -d $OUT/${TAG}_k2048_pmc_mfma -- python3 tools/fused_pmc.py > /dev/null 2>&1 || echo "k2048 mfma pass failed"
python tools/pmc_summarize.py --commit "$COMMIT" --csrc_sha "$SHA" --device "$DEV" $OUT/${TAG}_cfg5_pmc_FETCH_SIZE $OUT/${TAG}_cfg5_pmc_WRITE_SIZE $OUT/${TAG}_cfg5_pmc_mfma > $OUT/${TAG}_pmc_cfg5.json
# round 4: the same planner on the persistent step-chain kernel (option dehoog_chain = 1)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_chain_pmc_$C -- python3 tools/cfg5_chain_pmc.py > /dev/null 2>&1 || echo "chain pmc $C failed"
done
timeout -k 10 200 rocprofv3 --pmc $PMC_MFMA --kernel-trace --output-format csv -d $OUT/${TAG}_chain_pmc_mfma -- python3 tools/cfg5_chain_pmc.py > /dev/null 2>&1 || echo "chain mfma pass failed"
python tools/pmc_summarize.py --commit "$COMMIT" --csrc_sha "$SHA" --device "$DEV" $OUT/${TAG}_chain_pmc_FETCH_SIZE $OUT/${TAG}_chain_pmc_WRITE_SIZE $OUT/${TAG}_chain_pmc_mfma > $OUT/${TAG}_pmc_cfg5_chain.json
python tools/pmc_summarize.py --commit "$COMMIT" --csrc_sha "$SHA" --device "$DEV" $OUT/${TAG}_k2048_pmc_FETCH_SIZE $OUT/${TAG}_k2048_pmc_WRITE_SIZE $OUT/${TAG}_k2048_pmc_mfma > $OUT/${TAG}_pmc_k2048.json
timeout -k 10 120 python tools/cfg5_breakdown.py > $OUT/${TAG}_cfg5_breakdown.txt 2>/dev/null
timeout -k 10 200 python tools/fused_probe.py 2048 4096 > $OUT/${TAG}_fused_probe.jsonl 2>/dev/null
timeout -k 10 300 python tools/configs_bench.py > $OUT/${TAG}_configs.json 2>/dev/null
cut -c1-600 $OUT/${TAG}_bench_line.json
head -12 $OUT/${TAG}_bench_kernel_stats.csv
cat $OUT/${TAG}_cfg5_breakdown.txt
