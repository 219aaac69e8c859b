#!/bin/bash
# developer script: kernel time (rocprofv3 kernel stats) of ONE DepthConvBlock without adaptor at 68x120, widths 256 / 384: ring tail with
# the head inside (DCVC_T32H=1, default) against head launch + ring tail (DCVC_T32H=0)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 256 384; do
  for v in 1 0; do
    export DCVC_T32H=$v
    rm -rf gpurun_out/abh
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abh -- python3 tools/kbench.py chain $c 1 68 120 > /dev/null 2>&1
    f=$(find gpurun_out/abh -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$c" "$v" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dcb_" in r["Name"]]
tot = 0.0
for r in rows:
    tot += float(r["AverageNs"]) / 1e3
    print("C=%s T32H=%s  %-66s calls %s  mean %.1f us" % (sys.argv[2], sys.argv[3], r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3))
print("C=%s T32H=%s  block total %.1f us" % (sys.argv[2], sys.argv[3], tot))
PY
  done
done
rm -rf gpurun_out/abh
