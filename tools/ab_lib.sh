#!/bin/bash
# developer script: A/B of two builds of the library on one box - tools/kbench.py shapes, alternating
#   tools/ab_lib.sh libdcvc_amd_prev.so [shape ...]     (shapes as "C H W"; default: the frame's main ones)
PREV=${1:-libdcvc_amd_prev.so}
shift
SHAPES=("$@")
[ ${#SHAPES[@]} -eq 0 ] && SHAPES=("256 136 240" "320 136 240" "384 136 240" "256 68 120" "384 68 120" "512 68 120")
for r in 1 2 3; do
  for s in "${SHAPES[@]}"; do
    echo "new  $(python tools/kbench.py $s 2>&1 | grep -v amdgpu.ids)"
    echo "prev $(DCVC_AMD_LIB=$PREV python tools/kbench.py $s 2>&1 | grep -v amdgpu.ids)"
  done
done
