#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
timeout -k 10 200 python tools/dehoog_bwd_bench.py 16384 655360 > $O/new.json 2> $O/new.err; echo rc=$?; tail -5 $O/new.err; cat $O/new.json | cut -c1-400
