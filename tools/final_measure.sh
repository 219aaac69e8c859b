# Developer script: the measurements kept under profiles/ (run on the GPU box from the repo root).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
rm -rf $O && mkdir -p $O
python3 bench.py > $O/r02_bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/r02_bench_steps20.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roof -- python3 bench.py --roofline-only > $O/r02_roofline_under_rocprof.json 2>> $O/bench.err
cp $(find $O/prof_roof -name "*kernel_stats.csv" | head -1) $O/r02_roofline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline > $O/r02_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/r02_bench_kernel_stats.csv
rm -rf $O/prof_roof $O/prof_bench
python3 bench.py --frame 3840x2160 --no-cpu-baseline > $O/r02_bench_4k.json 2>> $O/bench.err
python3 tools/iframe_time.py > $O/r02_iframe_time.txt 2>> $O/bench.err
python3 tools/kbench.py > $O/r02_kbench.txt 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl4 -- python3 tools/count_launches.py 4 > /dev/null 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl12 -- python3 tools/count_launches.py 12 > /dev/null 2>> $O/bench.err
python3 tools/count_launches.py --diff $O/cl4 $O/cl12 8 > $O/r02_launches_per_pair.txt
rm -rf $O/cl4 $O/cl12
echo done
