#!/bin/bash
# round-3 measurement set: launches per P-frame pair, bench line, roofline kernel stats (rocprofv3)
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl4 -- python3 $GRAFT_REPO_ROOT/tools/count_launches.py 4 > $O/cl4.log 2>&1 || { tail -5 $O/cl4.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl12 -- python3 $GRAFT_REPO_ROOT/tools/count_launches.py 12 > $O/cl12.log 2>&1 || { tail -5 $O/cl12.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/count_launches.py --diff gpurun_out/cl4 gpurun_out/cl12 8 | tee gpurun_out/r03_launches_per_pair.txt
python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { tail -5 gpurun_out/r03_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r03_bench.json'))
print('value', d['value'], 'enc', d['enc_fps_per_gpu'], 'dec', d['dec_fps_per_gpu'], 'roofline', d['roofline']['frac'], d['roofline']['kernel_ms'])"
