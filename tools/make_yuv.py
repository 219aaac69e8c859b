"""Developer tool: writes N frames of the synthetic sequence (opendcvc_amd.weights.synthetic_frame_yuv420, the SURVEY 8d recipe)
as a planar 8-bit YUV 4:2:0 file, the input format of opendcvc_amd.harness / the reference's test_video.py.
    python tools/make_yuv.py WIDTH HEIGHT FRAMES SEED OUT.yuv"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opendcvc_amd import weights  # noqa: E402

w, h, n, seed, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
with open(out, "wb") as f:
    for fi in range(n):
        for plane in weights.synthetic_frame_yuv420(h, w, fi, seed):
            f.write(plane.tobytes())
print(out, os.path.getsize(out), "bytes")
