#!/bin/bash
# developer script: layer tests + timing of the large-map heads, 128-pixel form on and off (one box)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_layers.py -x -q -k "adaptor_block_large or head128 or large_map or chain or depth_conv_block" > gpurun_out/h128_tests.log 2>&1 || { tail -30 gpurun_out/h128_tests.log; exit 1; }
tail -2 gpurun_out/h128_tests.log
for v in 1 0 1 0; do
  DCVC_H128=$v python tools/kbench.py adapt 2>&1 | grep -v amdgpu.ids | sed "s/^/H128=$v /"
  for C in 256 320 384; do DCVC_H128=$v python tools/kbench.py $C 136 240 2>&1 | grep -v amdgpu.ids | sed "s/^/H128=$v /"; done
done | tee gpurun_out/h128_kbench.log
