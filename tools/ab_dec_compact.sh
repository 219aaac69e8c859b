#!/bin/bash
# developer script: A/B of the decoder hand-off on one box - compacted on the device (default) against whole-array copies
# (DCVC_DEC_COMPACT=0): sequential decoder profile, then bench.py --steps 20 twice each, alternating
mkdir -p gpurun_out
for v in 1 0; do
  echo "== DCVC_DEC_COMPACT=$v tools/dec_profile.py"
  DCVC_DEC_COMPACT=$v python tools/dec_profile.py 2>&1 | grep -v amdgpu.ids | head -12
done
for r in 1 2; do
  for v in 1 0; do
    DCVC_DEC_COMPACT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-mode > gpurun_out/r4_deccompact_${v}_$r.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_deccompact_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["gop_weighted_value"], d["enc_fps_per_gpu"], d["dec_fps_per_gpu"])
PY
