#!/bin/bash
# developer script: kernel time (rocprofv3 kernel stats) of one width-128 DepthConvBlock at the hyper path's map sizes, ring tail with
# the head inside (DCVC_T32_128=1, default) against dcb_tail_kernel<..., HEADIN> (DCVC_T32_128=0)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for hw in "17 30" "34 60" "68 120"; do
  for v in 1 0; do
    export DCVC_T32_128=$v
    rm -rf gpurun_out/abc
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abc -- python3 tools/kbench.py chain 128 1 $hw > /dev/null 2>&1
    f=$(find gpurun_out/abc -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$hw" "$v" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dcb_tail" in r["Name"]]
for r in rows:
    print("%s T32_128=%s  %-60s calls %s  mean %.1f us" % (sys.argv[2], sys.argv[3], r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
rm -rf gpurun_out/abc
