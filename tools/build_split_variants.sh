#!/bin/bash
# tools only: builds of the library that differ in the compile-time knobs of the latency-split bodies (nlc_rollout.h:
# NLC_SPLIT_PREFETCH, NLC_SPLIT_COST_WAVE), as tools/_ab/libnlc_<name>.so, for same-box A/Bs (tools/split_ab.py).
#   tools/build_split_variants.sh name1:"-DFLAG=1 -DOTHER=2" name2:"..."
set -e
cd "$(dirname "$0")/../neurallaplacecontrol_amd/csrc"
TUS="kernels_nl kernels_fused kernels_nl_rep"
CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value"
mkdir -p ../../tools/_ab
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  mkdir -p build_ab/$name
  pids=""
  for tu in $TUS; do
    /opt/rocm/bin/hipcc $CXXFLAGS $flags -c $tu.hip -o build_ab/$name/$tu.o 2> build_ab/$name/$tu.log &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  objs=$(ls build/*.o | grep -v "_phase.o" | grep -v -E "build/(kernels_nl|kernels_fused|kernels_nl_rep|kernels_nl_pd2|kernels_nl_sp1)\.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/libnlc_$name.so $objs build_ab/$name/kernels_nl.o build_ab/$name/kernels_fused.o build_ab/$name/kernels_nl_rep.o -ldl
  echo "built tools/_ab/libnlc_$name.so"
done
