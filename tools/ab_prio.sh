#!/bin/bash
# developer script: kbench A/B of the FFN-priority experiment build (-DT128_FFN_PRIO -> libdcvc_amd_prio.so) against the shipped one
mkdir -p gpurun_out/prio
for rep in 1 2 3; do
  for s in "256 136 240" "320 136 240" "384 136 240" "256 68 120" "384 68 120" "512 68 120"; do
    for v in libdcvc_amd.so libdcvc_amd_prio.so; do
      DCVC_AMD_LIB=$v python3 tools/kbench.py $s 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
    done
  done
done > gpurun_out/prio/ab.txt 2>&1
timeout -k 10 120 tools/mb/bin/coissue_mb > gpurun_out/prio/coissue_r2_again.txt 2>&1
