"""Golden vectors for the harness log, generated in the BUILD container by importing the
reference's own functions (src/utils/common.py generate_log_json, src/utils/metrics.py calc_psnr,
test_video.py qp spacing).  Output: tests/golden/harness_log.json (inputs + expected outputs, data only).

    python tests/golden/make_golden_harness.py
"""
import json
import os
import sys

import numpy as np

REF = os.environ.get("DCVC_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from src.utils.common import generate_log_json  # noqa: E402
from src.utils.metrics import calc_psnr  # noqa: E402

rng = np.random.default_rng(77)
cases = []
for name, n, comps, verbose, times, all_i in (("yuv_verbose", 9, 4, True, None, False), ("yuv_times", 14, 4, False, (0.0123, 0.0456), False),
                                            ("single", 6, 1, True, None, False), ("all_intra", 4, 4, False, None, True)):
    types = [0 if (i == 0 or all_i or (name == "yuv_times" and i % 7 == 0)) else 1 for i in range(n)]
    bits = [int(v) for v in rng.integers(800, 90000, n)]
    psnrs = [[float(v) for v in rng.uniform(28, 45, comps)] for _ in range(n)]
    ssims = [[float(v) for v in rng.uniform(0.9, 1.0, comps)] for _ in range(n)]
    kw = {} if times is None else dict(avg_encoding_time=times[0], avg_decoding_time=times[1])
    out = generate_log_json(n, 1920 * 1080, 12.5, types, bits, psnrs, ssims, verbose=verbose, **kw)
    cases.append(dict(name=name, frame_pixel_num=1920 * 1080, test_time=12.5, frame_types=types, bits=bits, psnrs=psnrs,
                      ssims=ssims, verbose=verbose, times=times,
                      expect=[[k, (v if not isinstance(v, list) else [float(x) for x in v])] for k, v in out.items()]))
psnr_kat = []
for shape, noise in (((16, 24), 3.0), ((8, 8), 0.0), ((4, 4), 1e-7), ((5, 7), 300.0)):
    a = rng.integers(0, 256, shape).astype(np.uint8)
    b = a.astype(np.float32) + rng.normal(0, noise, shape).astype(np.float32) if noise else a.astype(np.float32)
    psnr_kat.append(dict(a=a.tolist(), b=[[float(v) for v in r] for r in b], psnr=float(calc_psnr(a, b))))
qps = {str(r): [int(i + 0.5) for i in np.linspace(0, 63, num=r)] for r in (2, 4, 6, 64)}     # test_video.py:452-455
json.dump(dict(cases=cases, psnr=psnr_kat, sweep_qps=qps), open(os.path.join(os.path.dirname(__file__), "harness_log.json"), "w"))
print("cases", len(cases), "psnr", [k["psnr"] for k in psnr_kat])
