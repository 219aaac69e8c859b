#!/bin/bash
# developer script: the 32-pixel tails with ring depth 12 (libdcvc_amd_g32d12.so) and, on top, two workgroups per CU
# (256 registers: libdcvc_amd_g32w2.so) against the shipped build (ring 24, one workgroup per CU)
mkdir -p gpurun_out/g32w2
LIBS="libdcvc_amd.so libdcvc_amd_g32d12.so libdcvc_amd_g32w2.so"
for rep in 1 2 3; do
  for s in "256 68 120" "384 68 120" "512 68 120" "128 34 60"; do
    for v in $LIBS; do
      DCVC_AMD_LIB=$v python3 tools/kbench.py $s 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
    done
  done
done > gpurun_out/g32w2/kbench.txt 2>&1
for rep in 1 2 3; do
  for v in $LIBS; do
    echo "$v $(DCVC_AMD_LIB=$v python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-exact-mode 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "enc", d["enc_fps_per_gpu"], "dec", d["dec_fps_per_gpu"], "bpp", d["gop_bpp"], "psnr", d["psnr"]["weighted_6y_u_v"])')"
  done
done > gpurun_out/g32w2/bench.txt 2>&1
