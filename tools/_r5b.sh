#!/bin/bash
# round 5, second GPU call: new tests (generic shapes, sharded C client, independent processes, recognition), phase clocks of the
# latency-split bodies, the fused body under sharing with longer runs
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_planner.py tests/test_gpu_batched_sharded.py -x -q -m gpu -k "shapes_without or sharded_protocol or independent_planner or literal or closure or c_client or timeout" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout -k 10 300 python tools/split_phase_clocks.py > $O/split_phase_clocks.json 2> $O/split_phase_clocks.err; echo "phase clocks rc=$?"; tail -3 $O/split_phase_clocks.err
timeout -k 10 400 python tools/fused_sharing.py --procs 1,3,6 --bodies auto,2 --commands 3000 > $O/fused_sharing.json 2> $O/fused_sharing.err; echo "sharing rc=$?"; grep -v amdgpu.ids $O/fused_sharing.err | tail -8
timeout -k 10 300 python tools/fused_sharing.py --procs 6 --bodies auto,2 --commands 3000 --host-spin 2 > $O/fused_sharing_spin2.json 2> $O/fused_sharing_spin2.err; echo "sharing spin2 rc=$?"; grep -v amdgpu.ids $O/fused_sharing_spin2.err | tail -4
