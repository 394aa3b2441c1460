#!/bin/bash
# tools/ilt_depth_ab.sh <experiments library>: the row-per-lane Fourier ILT kernel with one and two tiles in flight per wavefront
LIB=$1
for rep in 1 2 3; do
  for depth in 1 2; do
    for dbg in 0 1; do
      echo -n "depth=$depth dbg=$dbg: "
      NLC_ILT_DEPTH=$depth NLC_ILT_DBG=$dbg timeout -k 10 120 python tools/ilt_only.py 655360 fourier 17 $LIB 2>/dev/null | tr '\n' ' ' || exit 1
      echo
    done
  done
done
