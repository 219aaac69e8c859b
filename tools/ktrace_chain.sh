#!/bin/bash
# developer script: kernel-trace durations of the kbench chain (C=384, 68x120, 4 blocks)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/kt2
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 ${PROBE:-tools/kbench.py} ${MODE:-chain 384 4} > $O/out.txt 2> $O/err.txt
cat $O/out.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt2/p/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prev = None
for r in rows[-26:]:
    n = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev) / 1e3 if prev else 0
    prev = int(r['End_Timestamp'])
    print('%-60s %7.1f us gap %6.1f' % (n[-60:], d, gap))
PY
rm -rf $O/p
