#!/bin/bash
# developer script (round 4): decoder hand-off without copy commands - parity tests, launches per P-frame pair, decoder host profile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4s1
mkdir -p $O
python -m pytest tests/test_gpu_codec.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl4 -- python3 tools/count_launches.py 4 > /dev/null 2>> $O/err.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl12 -- python3 tools/count_launches.py 12 > /dev/null 2>> $O/err.log &&
python3 tools/count_launches.py --diff $O/cl4 $O/cl12 8 > $O/launches_per_pair.txt
rm -rf $O/cl4 $O/cl12
head -3 $O/launches_per_pair.txt
python3 tools/dec_profile.py > $O/dec_profile.txt 2>&1
head -3 $O/dec_profile.txt
