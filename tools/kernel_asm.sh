#!/bin/bash
# Usage: tools/kernel_asm.sh <file.hip> <kernel-name-substring> : compiles the device code of one source file to
# assembly, prints the resource usage of matching kernels and the wait/branch/barrier skeleton of the first match.
set -e
SRC=/root/repo/neurallaplacecontrol_amd/csrc/$1
EXTRA=""
[ "$1" = "kernels_mppi.hip" ] && EXTRA="-ffp-contract=off"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only $EXTRA -S -o /tmp/kasm.s "$SRC" 2>&1 | grep -v hip-link || true
grep -E "^\s+\.(vgpr_count|name|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" /tmp/kasm.s | paste - - - - - | sed 's/ \+/ /g; s/\t/ /g' | grep -i "$2" || true
