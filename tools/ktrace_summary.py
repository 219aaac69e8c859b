"""Developer tool: per-kernel GPU time of the last N P-frame pairs in a rocprofv3 --kernel-trace CSV of
tools/seq_run.py (kernels per pair counted from the trace: pairs are separated by the I frame / warm-up)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per_pair = int(sys.argv[3])
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']),
             int(r['Workgroup_Size_X'])) for r in rows)
last = ev[-per_pair * npairs:]
agg = collections.OrderedDict()
for s, e, n, g, wg in last:
    key = (re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)[:44], g, wg)
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{k[0]:46s} grid={k[1]:5d} wg={k[2]:4d}  n/pair={v[0]/npairs:5.1f}  avg={v[1]/v[0]/1e3:7.1f}us  per pair={v[1]/npairs/1e3:7.1f}us {100*v[1]/tot:5.1f}%")
print('kernels', len(ev), 'total per pair', tot / npairs / 1e3, 'us')
