#!/bin/bash
# round 5, first GPU call: the sharded / fused-body tests, the bench lines, the fused body under GPU sharing
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_batched_sharded.py tests/test_gpu_planner_bodies.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench_line.json 2> $O/bench.err && cut -c1-300 $O/bench_line.json
for c in 0 4 d4; do timeout -k 10 200 python bench.py --config $c --steps 20 --no-ilt > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "cfg $c rc=$?"; cut -c1-200 $O/bench_cfg$c.json; done
timeout -k 10 400 python tools/fused_sharing.py --procs 1,3,6 --bodies auto,2,3 --commands 300 > $O/fused_sharing.json 2> $O/fused_sharing.err; echo "sharing rc=$?"; cat $O/fused_sharing.err | tail -12
