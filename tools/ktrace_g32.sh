#!/bin/bash
# developer script: per-launch durations of the small-map tails inside real frames (kernel trace of 1 I + 6 P frames)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/kt
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 tools/count_launches.py 6 > /dev/null 2> $O/err.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt/p/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp'])) for r in rows]
# last 120 launches: kernel short name + duration + gap to previous
out = []
prev_end = None
for r in rows:
    n = r['Kernel_Name']
    short = ('T32<' + n.split('kernel<')[1][:3] + '>') if 'tail128' in n and 'G32' in n else ('T128<' + n.split('kernel<')[1][:3] + '>') if 'tail128' in n else n.split('(')[0][-28:]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gap = (int(r['Start_Timestamp']) - prev_end) / 1e3 if prev_end else 0
    prev_end = int(r['End_Timestamp'])
    out.append('%-30s %7.1f us  gap %6.1f' % (short, d, gap))
open('gpurun_out/kt/tail.txt', 'w').write('\n'.join(out))
g = [d for n, d, _ in sel if 'tail128' in n and 'G32' in n and '384' in n.split('kernel<')[1][:4]]
print('G32<384> launches', len(g), 'min %.1f med %.1f max %.1f' % (min(g), sorted(g)[len(g)//2], max(g)))
print(' '.join('%.0f' % d for d in g[-40:]))
PY
rm -rf $O/p
