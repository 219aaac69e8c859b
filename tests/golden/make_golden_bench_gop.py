"""The bench's own workload from the reference (BASELINE.json configs[1]: one 32-frame GOP of 1080p YUV 4:2:0, qp 32): the
REFERENCE's DMCI / DMC in .half() on the CPU (the arithmetic its published fps numbers were measured in; torch fallback ops, its own
rANS coder) on the frames bench.py codes - opendcvc_amd.weights.synthetic_frame_yuv420(1080, 1920, fi, seed 0), prepared and
scored with the reference harness's own functions (ycbcr420_to_444_np, np_image_to_tensor, the fp16 cast, replicate_pad,
get_distortion) and driven like test_video.py:164-214,258-285 (intra period 32, feature refresh at frame 1, two coders).
Output: tests/golden/bench_gop_f16.json: per frame bytes, sha256, PSNR (weighted, y, u, v), and the GOP's bpp / mean PSNR.
About 5 minutes.  Build container only.

    python tests/golden/make_golden_bench_gop.py
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

import make_golden as G  # noqa: E402  (load_models, sha, INDEX_MAP)
import ref_harness  # noqa: E402

H, W, GOP, QP, SEED_SRC = 1080, 1920, 32, 32, 0


def main():
    DMC, DMCI, *_ = ref_harness.load()
    import test_video as tv
    from src.layers.cuda_inference import replicate_pad
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    i_net, p_net = G.load_models(DMC, DMCI)
    i_net.half()
    p_net.half()
    for m in (i_net, p_net):
        m.set_use_two_entropy_coders(True)
    pr, pb = (-W) % 16, (-H) % 16
    args = dict(src_type="yuv420", calc_ssim=False)
    frames, streams = [], []
    p_net.set_curr_poc(0)
    t0 = time.time()
    last_qp = 0

    def planes(fi):
        y, u, v = G.weights.synthetic_frame_yuv420(H, W, fi, SEED_SRC)
        return y, u, v

    def model_input(y, u, v):       # test_video.py:74-91 + :179
        x = tv.np_image_to_tensor(tv.ycbcr420_to_444_np(y[None], np.stack([u, v])), "cpu").to(torch.float16)
        return replicate_pad(x, pb, pr)

    for fi in range(GOP):
        x = model_input(*planes(fi))
        use_ada_i = 0
        if fi == 0:
            cur = QP
            enc = i_net.compress(x, QP)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, enc["x_hat"])
        else:
            if fi % 32 == 1:
                use_ada_i = 1
                p_net.prepare_feature_adaptor_i(last_qp)
            cur = p_net.shift_qp(QP, G.INDEX_MAP[fi % 8])
            enc = p_net.compress(x, cur)
            last_qp = cur
        streams.append((fi == 0, cur, use_ada_i, enc["bit_stream"]))
        frames.append(dict(type="I" if fi == 0 else "P", qp=cur, use_ada_i=use_ada_i, bytes=len(enc["bit_stream"]),
                           sha256=G.sha(enc["bit_stream"])))
        print("encoded", fi, len(enc["bit_stream"]), round(time.time() - t0), "s", flush=True)
    p_net.set_curr_poc(0)
    for fi, (is_i, cur, use_ada_i, bits) in enumerate(streams):
        sps = dict(height=H + pb, width=W + pr, ec_part=1, use_ada_i=use_ada_i)
        if is_i:
            dec = i_net.decompress(bits, sps, cur)
            p_net.clear_dpb()
            p_net.add_ref_frame(None, dec["x_hat"])
        else:
            if use_ada_i:
                p_net.reset_ref_feature()
            dec = p_net.decompress(bits, sps, cur)
        y, u, v = planes(fi)
        psnr, _ = tv.get_distortion(args, dec["x_hat"][:, :, :H, :W], y, u, v, None)
        frames[fi]["psnr"] = [float(p) for p in psnr]
        print("decoded", fi, frames[fi]["psnr"][0], round(time.time() - t0), "s", flush=True)
    total = sum(f["bytes"] for f in frames)
    out = dict(height=H, width=W, gop=GOP, qp=QP, src_seed=SEED_SRC, seed=G.SEED, thres=G.THRES, frames=frames,
               gop_bpp=total * 8.0 / (GOP * H * W), psnr_mean=[float(v) for v in np.mean([f["psnr"] for f in frames], axis=0)],
               mode="reference .half() on the CPU", seconds=round(time.time() - t0, 1))
    json.dump(out, open(os.path.join(HERE, "bench_gop_f16.json"), "w"), indent=1)
    print("gop bpp", out["gop_bpp"], "psnr", out["psnr_mean"])


if __name__ == "__main__":
    main()
