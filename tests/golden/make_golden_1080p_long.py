"""A longer 1080p record from the reference (does parity hold along the temporal chain?): the REFERENCE's DMCI / DMC on the CPU
(torch fallback ops, its own rANS coder), padded 1088 x 1920, 9 frames (I + 8 P), qp 32 with the reference's per-frame offsets,
feature refresh every 4 frames (use_ada_i on frames 1 and 5), two coders - once in fp32 and once in .half() (the benchmarked
mode's arithmetic), driven like test_video.py:164-214,258-285.  Output: tests/golden/seq_1080p_long.json (bytes, sha256, PSNR
per frame and mode; data only).  A few minutes.  Build container only.

    python tests/golden/make_golden_1080p_long.py [WxH frames out.json]      (3840x2160 2 seq_2160p.json: the 4K record)
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import make_golden as G  # noqa: E402  (load_models, sha, SEED / THRES / INDEX_MAP)
import ref_harness  # noqa: E402

H, W, QP, N, RESET = 1088, 1920, 32, 9, 4
OUT = "seq_1080p_long.json"
if len(sys.argv) > 1:       # e.g. `3840x2160 2 seq_2160p.json`: BASELINE.json configs[3]'s size (no padding: 2160 = 16 x 135), I + P
    W, H = (int(v) for v in sys.argv[1].lower().split("x"))
    N, OUT = int(sys.argv[2]), sys.argv[3]


def run(i_net, p_net, half):
    frames = []
    for m in (i_net, p_net):
        m.set_use_two_entropy_coders(True)
    p_net.set_curr_poc(0)
    t0 = time.time()
    streams, last_qp = [], 0
    for fi in range(N):
        x = torch.from_numpy(G.weights.synthetic_frame_yuv444(H, W, fi, 0))
        x = x.half() if half else x
        use_ada_i = 0
        if fi == 0:
            cur = QP
            enc = i_net.compress(x, QP)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, enc["x_hat"])
        else:
            if fi % RESET == 1:
                use_ada_i = 1
                p_net.prepare_feature_adaptor_i(last_qp)
            cur = p_net.shift_qp(QP, G.INDEX_MAP[fi % 8])
            enc = p_net.compress(x, cur)
            last_qp = cur
        streams.append((fi == 0, cur, use_ada_i, enc["bit_stream"]))
        frames.append(dict(type="I" if fi == 0 else "P", qp=cur, use_ada_i=use_ada_i, bytes=len(enc["bit_stream"]),
                           sha256=G.sha(enc["bit_stream"])))
        print("half" if half else "fp32", "encoded", fi, len(enc["bit_stream"]), round(time.time() - t0), "s", flush=True)
    p_net.set_curr_poc(0)
    for fi, (is_i, cur, use_ada_i, bits) in enumerate(streams):
        sps = dict(height=H, width=W, ec_part=1, use_ada_i=use_ada_i)
        if is_i:
            dec = i_net.decompress(bits, sps, cur)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, dec["x_hat"])
        else:
            if use_ada_i:
                p_net.reset_ref_feature()
            dec = p_net.decompress(bits, sps, cur)
        xh = dec["x_hat"].float().numpy().astype(np.float64)
        x = G.weights.synthetic_frame_yuv444(H, W, fi, 0).astype(np.float64)
        frames[fi]["psnr"] = float(-10 * np.log10(np.mean((xh - x) ** 2)))
        print("half" if half else "fp32", "decoded", fi, frames[fi]["psnr"], round(time.time() - t0), "s", flush=True)
    return frames


def main():
    DMC, DMCI, *_ = ref_harness.load()
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    out = dict(h=H, w=W, qp=QP, two=1, reset_interval=RESET, seed=G.SEED, thres=G.THRES)
    i_net, p_net = G.load_models(DMC, DMCI)
    out["fp32"] = run(i_net, p_net, False)
    i_net, p_net = G.load_models(DMC, DMCI)          # fresh instances (the fp32 run left fp32 masks in the models' caches):
    i_net.half()                                     # update() in fp32, then .half(), like test_video.py:398-404
    p_net.half()
    out["fp16"] = run(i_net, p_net, True)
    json.dump(out, open(os.path.join(HERE, OUT), "w"), indent=1)


if __name__ == "__main__":
    main()
