#!/bin/bash
# developer script: last verification of the round (run on the GPU box from the repo root): GPU tests, smoke, default and driver-style bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final3
rm -rf $O && mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/r04_gpu_tests_final.txt 2>&1
echo "tests rc=$?" >> $O/r04_gpu_tests_final.txt
tail -3 $O/r04_gpu_tests_final.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt
python3 bench.py > $O/r04_bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 > $O/r04_bench_driver_style.json 2>> $O/bench.err
echo done
