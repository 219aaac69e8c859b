"""Summarises the rocprofv3 --pmc passes of tools/kbench.py 256 136 240 (separate runs per counter set, as
MI355X_MICROARCH.md prescribes: tools/final_measure.sh) for the dominant kernel:

    python tools/pmc_summarize.py <dir with pmc_*/ sub-directories> <out.json> <out.txt>

out.json: HBM traffic per launch (FETCH_SIZE x 2 - the gfx950 correction - + WRITE_SIZE) against the algorithmic bytes, read
by bench.py for roofline.traffic; out.txt: mean of every collected counter per kernel.
"""
import csv, glob, json, os, sys
from collections import defaultdict

root, out_json, out_txt = sys.argv[1:4]
C, H, W = 256, 136, 240
P = H * W
vals = defaultdict(list)
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        kind = "tail128" if "dcb_tail128_kernel" in k else "head" if "dcb_head_kernel" in k else "tail64" if "dcb_tail_kernel" in k else None
        if kind:
            vals[(kind, r["Counter_Name"])].append(float(r["Counter_Value"]))
mean = {k: (sum(v) / len(v), len(v)) for k, v in vals.items()}
with open(out_txt, "w") as f:
    f.write("== python3 tools/kbench.py 256 136 240 under rocprofv3 --pmc <set> (one pass per set); mean per dispatch\n")
    for (kind, name), (m, n) in sorted(mean.items()):
        f.write("%-8s %-40s mean %16.1f  n=%d\n" % (kind, name, m, n))
    g = lambda name: mean.get(("tail128", name), (None, 0))[0]
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY"):
        f.write("tail128: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f\n" % (g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY"):
        f.write("tail128: SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f\n" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")))
    if g("TCP_TCC_READ_REQ_sum") and g("TCP_TCC_READ_REQ_LATENCY_sum"):
        f.write("tail128: L1->L2 read requests %.2f M = %.0f MB; mean latency %.0f cycles\n" %
                (g("TCP_TCC_READ_REQ_sum") / 1e6, g("TCP_TCC_READ_REQ_sum") * 128 / 1e6, g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")))
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        f.write("tail128: L2 hit rate %.3f\n" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
    if g("GRBM_GUI_ACTIVE") and g("TA_BUSY_avr"):
        f.write("tail128: TA busy %.3f of the kernel's cycles (GRBM_GUI_ACTIVE / 8 per XCD)\n" % (g("TA_BUSY_avr") / (g("GRBM_GUI_ACTIVE") / 8)))
f_tail, n = mean.get(("tail128", "FETCH_SIZE"), (None, 0))
w_tail, _ = mean.get(("tail128", "WRITE_SIZE"), (None, 0))
f_head, _ = mean.get(("head", "FETCH_SIZE"), (None, 0))
if f_tail is not None and w_tail is not None:
    fetch, write = 2 * f_tail * 1024, w_tail * 1024      # gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streams
    alg = 3 * P * C * 2 + 7 * C * C * 2 + 9 * C * 2       # a in, x' in, out; weights once
    res = {
        "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- python3 tools/kbench.py 256 136 240",
        "kernel": "t128::dcb_tail128_kernel<256>  (DepthConvBlock tail, C=256, 136x240, 255 workgroups of 512 threads)",
        "notes": ["FETCH_SIZE / WRITE_SIZE are KiB per dispatch (mean over %d dispatches)" % n,
                  "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streams -> doubled",
                  "calibration in the same run: dcb_head_kernel reads x (%.2f MB) + 0.13 MB weights; its 2 x FETCH_SIZE = %.1f MB"
                  % (P * C * 2 / 1e6, 2 * (f_head or 0) * 1024 / 1e6)],
        "raw_KiB": {"FETCH_SIZE_head": f_head, "FETCH_SIZE": f_tail, "dispatches": n, "WRITE_SIZE": w_tail},
        "hbm_bytes_per_launch": int(fetch + write), "fetch_bytes_corrected": int(fetch), "write_bytes": int(write),
        "algorithmic_bytes_per_launch": alg}
    json.dump(res, open(out_json, "w"), indent=1)
    print(json.dumps(res, indent=1))
print(open(out_txt).read())
