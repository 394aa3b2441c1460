#!/bin/bash
# timing variants of the int8-sliced encoder (tools only)
mkdir -p gpurun_out/r5r
for v in "" tools/_libnlc_i8nostage.so tools/_libnlc_i8nogates.so tools/_libnlc_i8noslice.so tools/_libnlc_i8nomfma.so tools/_libnlc_i8norec.so; do
  n=$(basename "${v:-product}" .so)
  timeout -k 10 200 python tools/i8_gemm_probe.py --steps 30 --windows 70000 ${v:+--lib $v} > gpurun_out/r5r/$n.txt 2> gpurun_out/r5r/$n.err || { echo "FAILED $n"; tail -3 gpurun_out/r5r/$n.err; exit 1; }
  echo "== $n $(grep -E '^i8 ' gpurun_out/r5r/$n.txt | grep -o '"gru_encode_kernel": [0-9.]*')"
done
