#!/bin/bash
# developer script: kernel timing of the large-map tails with several builds of the library (DCVC_AMD_LIB), interleaved
mkdir -p gpurun_out
for rep in 1 2 3; do
for v in ${LIBS:-libdcvc_amd_old.so libdcvc_amd.so}; do
  for C in ${@:-256}; do DCVC_AMD_LIB=$v python tools/kbench.py $C 136 240 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"; done
done; done | tee gpurun_out/ab_kbench.log
