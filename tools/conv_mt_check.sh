#!/bin/bash
# developer script: conv tests + the stand-alone convs of a 1080p frame, 128-pixel 3x3 kernel on / off
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_layers.py -x -q -k "conv" > gpurun_out/conv_tests.log 2>&1 || { tail -30 gpurun_out/conv_tests.log; exit 1; }
tail -2 gpurun_out/conv_tests.log
for v in 1 0 1; do
  DCVC_C128=$v python tools/kbench.py conv 2>&1 | grep "k3 s1\|s2" | sed "s/^/C128=$v /"
done | tee gpurun_out/conv_mt.log
