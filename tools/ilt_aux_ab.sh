#!/bin/bash
# tools/ilt_aux_ab.sh lib1.so lib2.so ...: the row-per-lane Fourier ILT kernel of several experiment builds on one box, full and
# memory-only (NLC_ILT_DBG=1), three interleaved rounds
for rep in 1 2 3; do
  for lib in "$@"; do
    for dbg in 0 1; do
      echo -n "$(basename $lib) dbg=$dbg: "
      NLC_ILT_ROWS=1 NLC_ILT_DBG=$dbg timeout -k 10 120 python tools/ilt_only.py 655360 fourier 17 $lib 2>/dev/null | tr '\n' ' ' || exit 1
      echo
    done
  done
done
