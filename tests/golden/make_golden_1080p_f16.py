"""The benchmarked mode at the benchmarked size, from the reference itself: the REFERENCE's DMCI / DMC in .half() on the CPU
(torch fallback ops - the arithmetic of its fp16 GPU path: every conv output and activation rounded to fp16 - and its own
rANS coder) on the padded 1088 x 1920 sequence of sequences.json (I, P, P; qp 32; two coders), driven like
test_video.py:164-214,258-285.  Output: tests/golden/seq_1080p_f16.json (bytes, sha256, PSNR per frame; data only).
Slow (half-precision convolutions on the CPU): ~20 - 40 minutes.  Build container only.

    python tests/golden/make_golden_1080p_f16.py
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import make_golden as G  # noqa: E402
import ref_harness  # noqa: E402


def main():
    DMC, DMCI, *_ = ref_harness.load()
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    i_net, p_net = G.load_models(DMC, DMCI)          # update() in fp32 first, like test_video.py:398-404 ...
    i_net.half()                                     # ... then .half()
    p_net.half()
    import numpy as np
    h, w, qp, n = 1088, 1920, 32, 3
    rec = dict(h=h, w=w, qp=qp, two=1, reset_interval=0, seed=G.SEED, thres=G.THRES, frames=[])
    for m in (i_net, p_net):
        m.set_use_two_entropy_coders(True)
    p_net.set_curr_poc(0)
    t0 = time.time()
    streams = []
    for fi in range(n):
        x = torch.from_numpy(G.weights.synthetic_frame_yuv444(h, w, fi, 0)).half()      # test_video.py:90
        if fi == 0:
            cur = qp
            enc = i_net.compress(x, qp)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, enc["x_hat"])
        else:
            cur = p_net.shift_qp(qp, G.INDEX_MAP[fi % 8])
            enc = p_net.compress(x, cur)
        streams.append((fi == 0, cur, enc["bit_stream"]))
        rec["frames"].append(dict(type="I" if fi == 0 else "P", qp=cur, use_ada_i=0, bytes=len(enc["bit_stream"]),
                                  sha256=G.sha(enc["bit_stream"])))
        print("encoded", fi, len(enc["bit_stream"]), round(time.time() - t0), "s", flush=True)
    p_net.set_curr_poc(0)
    for fi, (is_i, cur, bits) in enumerate(streams):
        sps = dict(height=h, width=w, ec_part=1, use_ada_i=0)
        if is_i:
            dec = i_net.decompress(bits, sps, cur)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, dec["x_hat"])
        else:
            dec = p_net.decompress(bits, sps, cur)
        xh = dec["x_hat"].float().numpy().astype(np.float64)
        x = G.weights.synthetic_frame_yuv444(h, w, fi, 0).astype(np.float64)
        rec["frames"][fi]["psnr"] = float(-10 * np.log10(np.mean((xh - x) ** 2)))
        print("decoded", fi, rec["frames"][fi]["psnr"], round(time.time() - t0), "s", flush=True)
    rec["mode"] = "reference .half() on the CPU"
    rec["seconds"] = round(time.time() - t0, 1)
    json.dump(rec, open(os.path.join(HERE, "seq_1080p_f16.json"), "w"), indent=1)
    print(json.dumps(rec["frames"], indent=1))


if __name__ == "__main__":
    main()
