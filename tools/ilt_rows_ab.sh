#!/bin/bash
# Same-box timing of the two Fourier ILT kernels (round 6): the term-per-lane stream (NLC_ILT_ROWS=0) against the row-per-lane
# kernel with direct global -> LDS loads, each also as memory-only (DBG 1) and arithmetic-only (DBG 2) variants, on an
# -DNLC_ILT_EXPERIMENTS=1 build of the library:  tools/ilt_rows_ab.sh <library.so> [N]
LIB=$1; N=${2:-655360}
for rep in 1 2; do
  for rows in 0 1; do
    for dbg in 0 1 2; do
      echo -n "rows=$rows dbg=$dbg: "
      NLC_ILT_ROWS=$rows NLC_ILT_DBG=$dbg timeout -k 10 120 python tools/ilt_only.py $N fourier 17 $LIB | tr '\n' ' ' || exit 1
      echo
    done
  done
done
echo -n "S=33 rows=0: "; NLC_ILT_ROWS=0 timeout -k 10 120 python tools/ilt_only.py $N fourier 33 $LIB | tr '\n' ' '; echo
echo -n "S=33 rows=1: "; NLC_ILT_ROWS=1 timeout -k 10 120 python tools/ilt_only.py $N fourier 33 $LIB | tr '\n' ' '; echo
echo -n "N=81920 rows=0: "; NLC_ILT_ROWS=0 timeout -k 10 120 python tools/ilt_only.py 81920 fourier 17 $LIB | tr '\n' ' '; echo
echo -n "N=81920 rows=1: "; NLC_ILT_ROWS=1 timeout -k 10 120 python tools/ilt_only.py 81920 fourier 17 $LIB | tr '\n' ' '; echo
