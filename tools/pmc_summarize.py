"""Summarises the two rocprofv3 --pmc passes of tools/kbench.py (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/pmc_dcb_tail.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/kbench.py 256 136 240
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/kbench.py 256 136 240
    python tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_dcb_tail.json
"""
import csv, glob, json, os, sys


def mean_counter(d, counter, needle):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch_dir, write_dir, out = sys.argv[1:4]
C, H, W = 256, 136, 240
P = H * W
f_tail, n = mean_counter(fetch_dir, "FETCH_SIZE", "dcb_tail_kernel")
f_head, _ = mean_counter(fetch_dir, "FETCH_SIZE", "dcb_head_kernel")
w_tail, _ = mean_counter(write_dir, "WRITE_SIZE", "dcb_tail_kernel")
w_head, _ = mean_counter(write_dir, "WRITE_SIZE", "dcb_head_kernel")
fetch = 2 * f_tail * 1024          # gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streams
write = w_tail * 1024
alg = 3 * P * C * 2 + 7 * C * C * 2 + 9 * C * 2      # a in, x' in, out; weights once
res = {
    "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- python3 tools/kbench.py 256 136 240",
    "kernel": "dcb_tail_kernel<_Float16, MT=4, NTW=4, NW=4>  (DepthConvBlock C=256, 136x240, 510 workgroups)",
    "notes": [
        "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (mean over %d dispatches)" % n,
        "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streams -> doubled",
        "calibration in the same run: dcb_head_kernel reads x (%.2f MB) + 0.13 MB weights; its 2 x FETCH_SIZE = %.1f MB"
        % (P * C * 2 / 1e6, 2 * f_head * 1024 / 1e6),
    ],
    "raw_KiB": {"FETCH_SIZE_head": f_head, "FETCH_SIZE": f_tail, "dispatches": n, "WRITE_SIZE_head": w_head, "WRITE_SIZE": w_tail},
    "hbm_bytes_per_launch": int(fetch + write),
    "fetch_bytes_corrected": int(fetch),
    "write_bytes": int(write),
    "algorithmic_bytes_per_launch": alg,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
