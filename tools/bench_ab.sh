#!/bin/bash
# tools/bench_ab.sh "<bench args>" lib1.so lib2.so ... : bench.py with each library build, interleaved 2x (same box)
ARGS=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    cp "$lib" neurallaplacecontrol_amd/libnlc_hip.so
    python bench.py $ARGS --no-cpu-baseline --no-ilt --no-other-configs --no-sliced-encoder 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels_avg_ms'].items()})" || exit 1
  done
done
