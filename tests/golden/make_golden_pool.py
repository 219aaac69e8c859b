"""Golden text for the merged log of a manifest run (test_video.py:517-532), written by the REFERENCE's own dump_json
(src/utils/common.py:49-60) in the build container.  Output: tests/golden/pool_log.json = {config, results, text}
(data only): `results` are per-job result dicts in arrival order, `text` is what the reference's main() writes for them.

    python tests/golden/make_golden_pool.py
"""
import io
import json
import os
import sys

REF = os.environ.get("DCVC_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
from src.utils.common import dump_json  # noqa: E402

config = {"root_path": "/data/", "test_classes": {
    "SetA": {"test": 1, "base_path": "a", "src_type": "yuv420", "sequences": {
        "one_64x64.yuv": {"width": 64, "height": 64, "frames": 5, "intra_period": -1},
        "two_96x64.yuv": {"width": 96, "height": 64, "frames": 4, "intra_period": 2}}},
    "Skipped": {"test": 0, "base_path": "s", "src_type": "yuv420", "sequences": {
        "no.yuv": {"width": 64, "height": 64, "frames": 9, "intra_period": -1}}},
    "SetB": {"test": 1, "base_path": "b", "src_type": "yuv420", "sequences": {
        "three_64x64.yuv": {"width": 64, "height": 64, "frames": 3, "intra_period": -1}}}}}
qps = [0, 63]
results = []
k = 0
for ds, seqs in (("SetB", ["three_64x64.yuv"]), ("SetA", ["two_96x64.yuv", "one_64x64.yuv"])):     # (arrival order is free)
    for seq in seqs:
        for rate_idx in (1, 0):
            k += 1
            results.append({"frame_pixel_num": 4096, "i_frame_num": 1, "p_frame_num": 3 + k, "ave_i_frame_bpp": 0.5 + k / 7,
                            "ave_i_frame_psnr": 30.0 + k / 3, "ave_all_frame_bpp": 1e-7 * k, "test_time": 12.0 * k,
                            "frame_bpp": [0.25 * k, 1 / 3], "ds_name": ds, "seq": seq, "rate_idx": rate_idx,
                            "qp_i": qps[rate_idx], "qp_p": qps[rate_idx]})
# what main() builds (test_video.py:517-528): datasets and sequences in manifest order, points keyed "%03d" in arrival order
log = {}
for ds in config["test_classes"]:
    if config["test_classes"][ds]["test"] == 0:
        continue
    log[ds] = {seq: {} for seq in config["test_classes"][ds]["sequences"]}
for res in results:
    log[res["ds_name"]][res["seq"]][f"{res['rate_idx']:03d}"] = res
buf = io.StringIO()
dump_json(log, buf, float_digits=6, indent=2)
json.dump(dict(config=config, qps=qps, results=results, text=buf.getvalue()),
          open(os.path.join(os.path.dirname(__file__), "pool_log.json"), "w"))
print(len(buf.getvalue()), "characters")
