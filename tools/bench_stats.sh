cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-exact-mode > $O/r03_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
rm -rf $O/prof_bench
head -8 $O/r03_bench_kernel_stats.csv | cut -c1-150
