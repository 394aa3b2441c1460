#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
timeout -k 10 300 python tools/split_phase_clocks.py rep > $O/split_phase_clocks_rep.json 2> $O/split_phase_clocks_rep.err; echo "phase clocks rc=$?"
b() { timeout -k 10 120 python bench.py --config 4 --steps 30 --no-ilt --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v['avg_ms'],4) for k,v in d['kernels_avg_ms'].items()})"; }
{ b; b --planner-opt dehoog_gru_chunks=2; b --planner-opt dehoog_gru_chunks=4; b --planner-opt dehoog_gru_chunks=8; b --planner-opt dehoog_gru_chunks=4 --planner-opt gru_coop=0; b --planner-opt dehoog_gru_chunks=4 --planner-opt dehoog_gru_lds_pad=0; b --planner-opt dehoog_gru_chunks=4 --planner-opt dehoog_gru_lds_pad=98304; b --planner-opt dehoog_gru_chunks=2 --planner-opt dehoog_streams=1; b --planner-opt dehoog_gru_chunks=4 --planner-opt dehoog_streams=1 --planner-opt dehoog_chain=0; } | tee $O/cfg5_gru_chunks.txt
timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; cut -c1-200 $O/bench_line.json
timeout -k 10 100 python bench.py --no-cpu-baseline --no-ilt --samples 2048 > $O/bench_2048.json 2>> $O/bench.err; cut -c1-200 $O/bench_2048.json
