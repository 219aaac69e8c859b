#!/bin/bash
# developer script: launches and kernel time per P-frame pair (the count_launches part of final_measure.sh);
# second pass with the 32-pixel ring tails switched off (DCVC_T32=0) on the same box
set -e
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final
mkdir -p $O
one() {
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl4 -- python3 tools/count_launches.py 4 > /dev/null 2>> $O/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl12 -- python3 tools/count_launches.py 12 > /dev/null 2>> $O/bench.err
  python3 tools/count_launches.py --diff $O/cl4 $O/cl12 8 > $1
  rm -rf $O/cl4 $O/cl12
  head -2 $1
}
one $O/${R}_launches_per_pair.txt
export DCVC_T32=0
one $O/${R}_launches_per_pair_t32_off.txt
