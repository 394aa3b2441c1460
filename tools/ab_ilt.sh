#!/bin/bash
# Same-box A/B timing of library builds: tools/ab_ilt.sh <N> lib1.so lib2.so ...  (each timed 3x, interleaved)
N=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    cp "$lib" neurallaplacecontrol_amd/libnlc_hip.so
    echo -n "$lib N=$N: "; timeout -k 10 120 python tools/ilt_only.py $N | head -1 || exit 1
  done
done
