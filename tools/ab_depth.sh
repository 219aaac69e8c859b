#!/bin/bash
# developer script: packets in flight between the encoder and the decoder stage (DCVC_PIPE_DEPTH), driver-style and default windows
mkdir -p gpurun_out/depth
for rep in 1 2 3; do
  for d in 4 1 2 8; do
    for k in 20 256; do
      w=5; [ $k = 256 ] && w=8
      echo "depth $d steps $k: $(DCVC_PIPE_DEPTH=$d python3 bench.py --steps $k --warmup $w --no-cpu-baseline --no-exact-mode 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "gop_weighted", d["gop_weighted_value"], d["gop_weighted_note"][-80:-22])')"
    done
  done
done > gpurun_out/depth/ab.txt 2>&1
