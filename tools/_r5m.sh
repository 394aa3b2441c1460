#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
for a in "2048 30000" "4096 10000" "1000 20000" "48 15000"; do timeout -k 10 280 python tools/fused_soak.py $a 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/fused_soak.txt
timeout -k 10 200 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_planner.py -x -q -m gpu -k "bench_other_configs or refresh_model" 2>&1 | tail -2
