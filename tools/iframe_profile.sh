#!/bin/bash
# developer script: kernel time of the I-frame codec by kernel (6 I frames; first one includes the graph captures)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/iprof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 tools/iframe_time.py > $O/iframe.txt 2> $O/err.txt
cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/iframe_kernel_stats.csv
rm -rf $O/p
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/iprof/iframe_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms over 6 I-frame pairs (+1 eager pass):', tot/1e6)
for r in rows[:25]:
    print('  %-90s calls %5s avg %8.1f us  %5.1f %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
