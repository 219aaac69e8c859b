#!/bin/bash
# developer script: the round's last measurement set after the compacted decoder hand-off (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final2
rm -rf $O && mkdir -p $O
python3 bench.py > $O/r04_bench.json 2> $O/bench.err
echo "bench done"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-mode > $O/r04_bench_steps20.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-mode > $O/r04_bench_steps20_b.json 2>> $O/bench.err
python3 bench.py --frame 3840x2160 --no-cpu-baseline --no-exact-mode > $O/r04_bench_4k.json 2>> $O/bench.err
echo "bench variants done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl4 -- python3 tools/count_launches.py 4 > /dev/null 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl12 -- python3 tools/count_launches.py 12 > /dev/null 2>> $O/bench.err
python3 tools/count_launches.py --diff $O/cl4 $O/cl12 8 > $O/r04_launches_per_pair_final.txt
rm -rf $O/cl4 $O/cl12
python3 tools/iframe_time.py > $O/r04_iframe_time.txt 2>> $O/bench.err
echo done
