#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 400 python tools/split_ab.py --rounds 2 tools/_ab/libnlc_base.so tools/_ab/libnlc_final.so > $O/split_ab_final.json 2> $O/split_ab_final.err; echo "ab rc=$?"; grep -v amdgpu.ids $O/split_ab_final.err | tail -6
timeout -k 10 300 python tools/split_phase_clocks.py rep split > $O/split_phase_clocks.json 2> $O/split_phase_clocks.err; echo "phase clocks rc=$?"
for n in 1 2 3 4; do timeout -k 10 120 python bench.py --config 4 --steps 30 --no-ilt --no-cpu-baseline --planner-opt dehoog_chain=0 --planner-opt dehoog_streams=$n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $n', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v['avg_ms'],4) for k,v in d['kernels_avg_ms'].items()})"; done | tee $O/cfg5_streams.txt
