"""Developer tool: how busy is the GPU inside the two-stage pipeline's timed window?

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 128 --no-cpu-baseline --no-exact-mode
    python3 tools/pipeline_busy.py gpurun_out/trace

Takes the two queues with the most kernels (the encoder's and the decoder's stream), finds the longest stretch in which
BOTH launch kernels without a pause of more than 30 ms (the pipelined window) and prints, for its middle 80 %: wall time, time
with at least one kernel running (union over all queues), time with kernels of both queues running at once, each queue's own
busy time, the sum of all kernel durations."""
import csv
import glob
import os
import sys
from collections import defaultdict


def merged(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def total(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            out += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def main(root):
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    per_q = defaultdict(list)
    for r in rows:
        per_q[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    top = sorted(per_q, key=lambda q: -len(per_q[q]))[:2]
    # stretches of each top queue without a 30 ms pause; the window = overlap of the two longest
    def longest(q):
        ks = sorted(per_q[q])
        best, cur0, last = (0, 0, 0), ks[0][0], ks[0][1]
        for s, e, _ in ks[1:]:
            if s - last > 30_000_000:
                if last - cur0 > best[0]:
                    best = (last - cur0, cur0, last)
                cur0 = s
            last = max(last, e)
        if last - cur0 > best[0]:
            best = (last - cur0, cur0, last)
        return best[1], best[2]
    (a0, a1), (b0, b1) = longest(top[0]), longest(top[1])
    w0, w1 = max(a0, b0), min(a1, b1)
    w0, w1 = w0 + (w1 - w0) // 10, w1 - (w1 - w0) // 10
    clip = lambda ks: [(max(s, w0), min(e, w1)) for s, e, *_ in ks if e > w0 and s < w1]
    allk = merged(clip([k for q in per_q for k in per_q[q]]))
    qa, qb = merged(clip(per_q[top[0]])), merged(clip(per_q[top[1]]))
    wall = w1 - w0
    ssum = sum(min(e, w1) - max(s, w0) for q in per_q for s, e, _ in per_q[q] if e > w0 and s < w1)
    print(f"pipelined window (middle 80 %): {wall / 1e6:.1f} ms")
    print(f"  some kernel running      {100 * total(allk) / wall:5.1f} %")
    print(f"  both streams at once     {100 * intersect(qa, qb) / wall:5.1f} %")
    print(f"  stream A busy            {100 * total(qa) / wall:5.1f} %   stream B busy {100 * total(qb) / wall:5.1f} %")
    print(f"  sum of kernel durations  {100 * ssum / wall:5.1f} % of the wall time")
    # idle gaps of the union
    gaps = [allk[i + 1][0] - allk[i][1] for i in range(len(allk) - 1)]
    hist = defaultdict(int)
    for g in gaps:
        hist[min(g // 20_000, 25)] += g
    print("  idle time by gap length (x20 us: ms): " + " ".join(f"{k}:{v / 1e6:.1f}" for k, v in sorted(hist.items())))
    # top kernels inside the window by total time
    byname = defaultdict(lambda: [0, 0])
    for q in per_q:
        for s, e, n in per_q[q]:
            if e > w0 and s < w1:
                byname[n][0] += 1
                byname[n][1] += e - s
    for n, (c, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {c:6d} x {t / c / 1e3:7.1f} us = {t / 1e6:7.1f} ms ({100 * t / wall:4.1f} % of wall)  {n[:100]}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace")
