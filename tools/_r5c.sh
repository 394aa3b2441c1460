#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout -k 10 1000 python tools/split_ab.py --rounds 2 tools/_ab/libnlc_base.so tools/_ab/libnlc_p2.so tools/_ab/libnlc_p3.so tools/_ab/libnlc_p4.so tools/_ab/libnlc_c3.so tools/_ab/libnlc_p3c3.so tools/_ab/libnlc_p4c3.so tools/_ab/libnlc_p6c3.so > $O/split_ab.json 2> $O/split_ab.err; echo "ab rc=$?"; grep -v amdgpu.ids $O/split_ab.err | tail -20
