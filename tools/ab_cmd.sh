#!/bin/bash
# Same-box A/B: tools/ab_cmd.sh "<command>" lib1.so lib2.so ...  -- runs <command> 3x per library, interleaved,
# with each library copied over neurallaplacecontrol_amd/libnlc_hip.so (restores the last one given at the end).
CMD=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    cp "$lib" neurallaplacecontrol_amd/libnlc_hip.so
    echo "== $lib"; bash -c "$CMD" || exit 1
  done
done
