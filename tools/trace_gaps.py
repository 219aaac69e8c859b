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


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace")
