#!/bin/bash
# developer script: layer tests + kernel timing of the large-map tails, 128-pixel form on and off (one box)
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_layers.py -x -q -k "large_map or chain" > gpurun_out/t128_tests.log 2>&1 || { tail -30 gpurun_out/t128_tests.log; exit 1; }
tail -2 gpurun_out/t128_tests.log
for C in ${@:-256}; do
for v in 1 0 1; do
  DCVC_T128=$v python tools/kbench.py $C 136 240 2>&1 | grep -v amdgpu.ids | sed "s/^/T128=$v /"
done
DCVC_AMD_DIAG=1 DCVC_STAMPS=1 python tools/kbench.py $C 136 240 2>&1 | grep stamps | tail -1
done | tee gpurun_out/t128_kbench.log
