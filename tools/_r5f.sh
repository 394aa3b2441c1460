#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
timeout -k 10 300 python tools/split_ab.py --rounds 3 --only cfg5 tools/_ab/libnlc_final.so tools/_ab/libnlc_final2.so > $O/split_ab_cfg5.json 2> $O/split_ab_cfg5.err; echo "ab rc=$?"; grep -v amdgpu.ids $O/split_ab_cfg5.err | tail -6
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_full.log; tail -4 $O/pytest_full.log
