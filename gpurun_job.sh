set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests6.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2_tests6.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_20.json 2> gpurun_out/r2_bench_20.err; echo "bench20 rc=$?"
python bench.py > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err; echo "bench rc=$?"
python tools/make_yuv.py 1920 1080 12 0 /tmp/seq1080.yuv && python -m opendcvc_amd.harness --src /tmp/seq1080.yuv --width 1920 --height 1080 --frames 12 --rate-num 4 --intra-period -1 --reset-interval 8 --verbose-json --out gpurun_out/r2_sweep_1080p_fp16.json 2> gpurun_out/r2_sweep.err; echo "sweep rc=$?"
