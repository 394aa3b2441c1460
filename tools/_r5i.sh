#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
# sizes that exercise ONE question each: 13000 points = 1016 blocks (one iteration, all full), 13120 = 1025 blocks (block 0 iterates twice)
timeout -k 10 120 python tools/_dhb_repro.py 12800 > $O/n12800.txt 2>&1; echo "12800 (1000 blocks) rc=$?"; grep -v amdgpu.ids $O/n12800.txt | tail -2 | cut -c1-200
