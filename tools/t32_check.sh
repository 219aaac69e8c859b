#!/bin/bash
# developer script: layer tests + kernel timing of the small-map tails, 32-pixel ring form on and off (one box)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_layers.py -x -q -k "depth_conv_block or chain or tail32" > gpurun_out/t32_tests.log 2>&1 || { tail -40 gpurun_out/t32_tests.log; exit 1; }
tail -2 gpurun_out/t32_tests.log
for v in 1 0 1; do
  for C in 256 384 512; do DCVC_T32=$v python tools/kbench.py $C 68 120 2>&1 | grep -v amdgpu.ids | sed "s/^/T32=$v /"; done
done | tee gpurun_out/t32_kbench.log
DCVC_AMD_DIAG=1 DCVC_STAMPS=1 python tools/kbench.py 384 68 120 2>&1 | grep stamps | tail -1 | tee -a gpurun_out/t32_kbench.log
for v in 1 0; do DCVC_T32=$v python tools/kbench.py chain 2>&1 | grep -v amdgpu.ids | sed "s/^/T32=$v /"; done | tee -a gpurun_out/t32_kbench.log
for v in 1 0; do DCVC_T32=$v python tools/kbench.py adapt 2>&1 | grep "68x120" | sed "s/^/T32=$v /"; done | tee -a gpurun_out/t32_kbench.log
