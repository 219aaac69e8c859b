set -o pipefail
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
python -m pytest tests/test_harness.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_20.json 2> gpurun_out/r2_bench_20.err; echo "bench20 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/bench -o bench -- python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/r2_bench_under_rocprof.json 2> gpurun_out/prof/bench.err; echo "rocprof bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/roof -o roof -- python bench.py --roofline-only > gpurun_out/r2_roofline_under_rocprof.json 2> gpurun_out/prof/roof.err; echo "rocprof roof rc=$?"
find gpurun_out/prof -name "*kernel_stats.csv" | head
rm -f gpurun_out/prof/*/*.db gpurun_out/prof/*/*kernel_trace.csv
python tools/iframe_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_iframe_time.txt; tail -2 gpurun_out/r2_iframe_time.txt
