"""Developer tool: idle time between consecutive kernels of one HIP stream in a `rocprofv3 --kernel-trace` CSV.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 64 --no-cpu-baseline
    python3 tools/trace_gaps.py gpurun_out/trace

Per queue: kernels, busy time, and the gaps between a kernel's end and the next one's start, split into short gaps
(< 25 us: dispatch / dependency latency between kernels of one captured run) and long ones (host work, waits)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + root)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    per_q = defaultdict(list)
    for r in rows:
        per_q[(r.get("Agent_Id"), r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    for q, ks in sorted(per_q.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        busy = sum(e - s for s, e, _ in ks)
        short, long_, nshort = 0, 0, 0
        hist = defaultdict(int)
        for (s0, e0, _), (s1, e1, _) in zip(ks, ks[1:]):
            g = s1 - e0
            if g < 0:
                continue
            if g < 25000:
                short += g
                nshort += 1
                hist[min(g // 1000, 24)] += 1
            else:
                long_ += g
        span = ks[-1][1] - ks[0][0]
        print(f"queue {q}: {len(ks)} kernels, span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), "
              f"short gaps {nshort} = {short/1e6:.2f} ms (mean {short/max(nshort,1)/1e3:.2f} us), long gaps {long_/1e6:.1f} ms")
        print("   short-gap histogram (us: count): " + " ".join(f"{k}:{v}" for k, v in sorted(hist.items())))


def union(root, lo_frac=0.0, hi_frac=1.0):
    """All queues together: fraction of the wall time during which at least one kernel runs, over the densest part of
    the trace (the longest stretch without an idle period > 20 ms = the bench's timed pipeline window)."""
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    iv = []
    for f in files:
        for r in csv.DictReader(open(f)):
            iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    iv.sort()
    # split into stretches separated by > 20 ms of nothing
    stretches, cur, end = [], [], None
    for s_, e_ in iv:
        if end is not None and s_ - end > 20_000_000:
            stretches.append(cur)
            cur = []
        cur.append((s_, e_))
        end = e_ if end is None else max(end, e_)
    stretches.append(cur)
    for st in sorted(stretches, key=lambda c: -(max(e for _, e in c) - c[0][0]))[:3]:
        t0, t1 = st[0][0], max(e for _, e in st)
        busy, idle_hist, last = 0, defaultdict(int), st[0][0]
        idle_total = 0
        cur_s, cur_e = st[0]
        for s_, e_ in st[1:]:
            if s_ > cur_e:
                busy += cur_e - cur_s
                g = s_ - cur_e
                idle_total += g
                idle_hist[min(g // 10_000, 50)] += g
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += cur_e - cur_s
        span = t1 - t0
        print(f"stretch of {span/1e6:.1f} ms, {len(st)} kernels: some kernel running {100*busy/span:.1f} % of the time, idle {idle_total/1e6:.1f} ms")
        print("   idle time by gap length (x10 us: ms): " + " ".join(f"{k}:{v/1e6:.1f}" for k, v in sorted(idle_hist.items())))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--union":
        union(sys.argv[1])
        sys.exit(0)
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace")
