"""Developer aid: output of one DepthConvBlock at a large map with the 128-pixel tail on / off (two processes), and where they differ."""
import os, subprocess, sys
import numpy as np

def child(path, C, H, W):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from opendcvc_amd import nn as L
    from tools.kbench import make_dcb_weights
    rng = np.random.default_rng(0)
    sd = make_dcb_weights(rng, "m", C, C, False)
    blk = L.DepthConvBlock(sd, "m", torch.float16)
    x = (torch.from_numpy(rng.standard_normal((H, W, blk.c_p)).astype(np.float32)) * 0.5).cuda().half()
    out = blk(x)
    torch.cuda.synchronize()
    np.save(path, out.float().cpu().numpy())

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    C, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 101, 123)
    outs = []
    for v in ("1", "0"):
        path = f"/tmp/t128_dbg_{v}.npy"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", path, str(C), str(H), str(W)],
                              env=dict(os.environ, DCVC_T128=v))
        outs.append(np.load(path))
    a, b = outs
    d = np.abs(a - b)
    rms = np.sqrt((b ** 2).mean())
    print(f"rms {rms:.4f} max|d| {d.max():.4f} mean|d| {d.mean():.6f}  nan {np.isnan(a).sum()}")
    bad = d > 0.02 * rms
    print("bad fraction", bad.mean())
    if bad.any():
        ys, xs, cs = np.nonzero(bad)
        print("bad by channel %32:", np.bincount(cs % 32, minlength=32))
        print("bad by channel //32:", np.bincount(cs // 32, minlength=C // 32))
        print("bad by (y%8):", np.bincount(ys % 8, minlength=8), " (x%16):", np.bincount(xs % 16, minlength=16))
        print("bad by tile x:", np.bincount(xs // 16), " tile y:", np.bincount(ys // 8))
        for k in range(min(8, len(ys))):
            print(ys[k], xs[k], cs[k], a[ys[k], xs[k], cs[k]], b[ys[k], xs[k], cs[k]])
