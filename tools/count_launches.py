"""Developer tool: codes 1 I + N P frames at 1080p (encoder and decoder, sequentially) so that two runs under
`rocprofv3 --kernel-trace --stats` with different N give the kernel launches per P-frame pair by difference:

    rocprofv3 --kernel-trace --stats --output-format csv -d out4 -- python3 tools/count_launches.py 4
    rocprofv3 --kernel-trace --stats --output-format csv -d out12 -- python3 tools/count_launches.py 12
    python3 tools/count_launches.py --diff out4 out12 8
"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


TIMES = {}


def calls(root):
    f = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)[0]
    ours, other = {}, {}
    for r in csv.DictReader(open(f)):
        mine = "_GLOBAL__N_1" in r["Name"] or "(anonymous namespace)" in r["Name"]      # (mangled or demangled: this library's kernels)
        (ours if mine else other)[r["Name"]] = int(r["Calls"])
        TIMES[(root, r["Name"])] = float(r["TotalDurationNs"])
    return ours, other


if len(sys.argv) > 1 and sys.argv[1] == "--diff":
    (a, ao), (b, bo), n = calls(sys.argv[2]), calls(sys.argv[3]), int(sys.argv[4])
    d = {k: (b.get(k, 0) - a.get(k, 0)) / n for k in b}
    do = {k: (bo.get(k, 0) - ao.get(k, 0)) / n for k in bo}
    print(f"this library's kernel launches per P-frame pair: {sum(d.values()):.1f}; torch / runtime kernels (copy commands, casts): {sum(do.values()):.1f}")
    us = lambda k: (TIMES.get((sys.argv[3], k), 0.0) - TIMES.get((sys.argv[2], k), 0.0)) / n / 1e3
    tot = sum(us(k) for k in list(d) + list(do))
    is_copy = lambda k: "rocclr" in k
    is_torch = lambda k: k.startswith("void at::") or "at::native" in k
    own = sum(us(k) for k in list(d) + list(do) if not is_copy(k) and not is_torch(k))
    print(f"kernel time per P-frame pair (encoder then decoder, one stream, nothing else on the GPU): {tot:.0f} us")
    print(f"  of which this library's kernels {own:.0f} us, runtime copy commands {sum(us(k) for k in do if is_copy(k)):.0f} us "
          f"(a difference of two processes: when one of them runs its copies at ~100 us each - seen in about half of the "
          f"processes, never in bench.py - this term is off, the kernels' is not), torch kernels {sum(us(k) for k in do if is_torch(k)):.0f} us")
    for k, v in sorted(d.items(), key=lambda kv: -us(kv[0])):
        if v:
            print(f"  {v:5.1f} x {us(k) / v:7.1f} us = {us(k):7.1f} us ({100 * us(k) / tot:4.1f} %)  {k[:90]}")
    for k, v in sorted(do.items(), key=lambda kv: -kv[1]):
        if v:
            print(f"  {v:5.1f} x {us(k) / v:7.1f} us = {us(k):7.1f} us ({100 * us(k) / tot:4.1f} %)  [other] {k[:80]}")
    sys.exit(0)

import torch
import bench
from opendcvc_amd.pipeline import SequenceDecoder, SequenceEncoder
torch.set_grad_enabled(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
(ie, pe), (idec, pdec) = bench.load_models(torch.float16, dev, 1, 0)
for m in (ie, pe, idec, pdec):
    m.set_use_two_entropy_coders(True)
frames = bench.make_frames(0, torch.float16, dev)[1][:n + 1]
enc = SequenceEncoder(ie, pe, 32, intra_period=64, defer_stream=True)
dec = SequenceDecoder(idec, pdec, 1080, 1920, True)
dec.defer = True
pkts = []
for x in frames:
    pkts += enc.encode(x)
pkts += enc.flush()
for p in pkts:
    dec.decode(p)
dec.flush()
torch.cuda.synchronize()
print("coded", len(pkts), "frames")
