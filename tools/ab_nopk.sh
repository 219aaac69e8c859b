#!/bin/bash
# developer script: A/B of a build without packed fp32 vector instructions (they serialise with MFMAs: tools/mb/coissue2_mb.hip)
#   tools/ab_nopk.sh [other lib]     (default libdcvc_amd_nopk.so, built with -Xclang -target-feature -Xclang -packed-fp32-ops)
OTHER=${1:-libdcvc_amd_nopk.so}
mkdir -p gpurun_out/nopk
for rep in 1 2 3; do
  for s in "256 136 240" "320 136 240" "384 136 240" "256 68 120" "384 68 120" "512 68 120" "128 34 60"; do
    for v in libdcvc_amd.so $OTHER; do
      DCVC_AMD_LIB=$v python3 tools/kbench.py $s 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
    done
  done
done > gpurun_out/nopk/ab.txt 2>&1
for rep in 1 2; do
  for v in libdcvc_amd.so $OTHER; do
    echo "$v $(DCVC_AMD_LIB=$v python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "enc", d["enc_fps_per_gpu"], "dec", d["dec_fps_per_gpu"], "roofline_ms", d["roofline"]["kernel_ms"], "exact", d.get("exact_mode", {}).get("enc_fps"), d.get("exact_mode", {}).get("dec_fps"), "bpp", d["gop_bpp"], "psnr", d["psnr"]["weighted_6y_u_v"])')"
  done
done > gpurun_out/nopk/bench.txt 2>&1
