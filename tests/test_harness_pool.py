"""Job fan-out of the harness (reference: test_video.py:381-442,472-532): manifest -> jobs, worker n on GPU n % gpus,
merged log in the reference's schema and number format.  CPU only: the codec is a stub (tests/stub_pool.py)."""
import io
import json
import os
import sys

import pytest

from opendcvc_amd import harness

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "pool_log.json")))


def test_merged_log_text_equals_the_reference_dump():
    """merge_results + dump_json reproduce, byte for byte, what the reference's main() writes (golden text produced by the
    reference's own dump_json: tests/golden/make_golden_pool.py)"""
    log = harness.merge_results(GOLD["config"], GOLD["results"])
    buf = io.StringIO()
    harness.dump_json(log, buf, float_digits=6, indent=2)
    assert buf.getvalue() == GOLD["text"]
    assert list(log) == ["SetA", "SetB"] and list(log["SetA"]) == ["one_64x64.yuv", "two_96x64.yuv"]


def test_jobs_follow_the_manifest_and_the_overrides():
    jobs = harness.jobs_from_config(GOLD["config"], dict(qp_i=[0, 63], force_root_path="/mnt/x"))
    assert [(j["ds_name"], j["seq"], j["rate_idx"], j["qp_i"]) for j in jobs] == [
        ("SetA", "one_64x64.yuv", 0, 0), ("SetA", "one_64x64.yuv", 1, 63), ("SetA", "two_96x64.yuv", 0, 0),
        ("SetA", "two_96x64.yuv", 1, 63), ("SetB", "three_64x64.yuv", 0, 0), ("SetB", "three_64x64.yuv", 1, 63)]
    assert jobs[2]["src_path"] == "/mnt/x/a/two_96x64.yuv" and jobs[2]["intra_period"] == 2 and jobs[2]["frame_num"] == 4
    assert jobs[0]["qp_p"] == 0 and jobs[0]["reset_interval"] == 32
    jobs = harness.jobs_from_config(GOLD["config"], dict(rate_num=4, force_frame_num=2, force_intra_period=8, qp_p=[1, 2, 3, 4]))
    assert [j["qp_i"] for j in jobs[:4]] == [0, 21, 42, 63] and [j["qp_p"] for j in jobs[:4]] == [1, 2, 3, 4]
    assert all(j["frame_num"] == 2 and j["intra_period"] == 8 for j in jobs)
    assert jobs[0]["src_path"] == "/data/a/one_64x64.yuv"
    assert all(j["src_type"] == "yuv420" for j in jobs)
    png = json.loads(json.dumps(GOLD["config"]))
    png["test_classes"]["SetA"]["src_type"] = "png"          # the reference's other source type: a directory of PNGs per sequence
    assert [j["src_type"] for j in harness.jobs_from_config(png, {})][:4] == ["png"] * 4
    bad = json.loads(json.dumps(GOLD["config"]))
    bad["test_classes"]["SetA"]["src_type"] = "rgb24"
    with pytest.raises(ValueError):
        harness.jobs_from_config(bad, {})


def test_worker_to_gpu_mapping_rule():
    assert [harness.worker_gpu(f"SpawnProcess-{n}", 8) for n in (1, 2, 8, 9, 16)] == [1, 2, 0, 1, 0]
    assert harness.worker_gpu("SpawnProcess-3", 0) == -1


def test_pool_of_spawned_workers_two_workers_two_gpus(tmp_path):
    """2 spawned workers x 2 'GPUs' with the stub codec: every job runs in a worker whose HIP_VISIBLE_DEVICES is its
    process number % 2, models are built once per worker, the merged log has the reference's layout."""
    sys.path.insert(0, HERE)
    opts = dict(qp_i=[0, 63], codec="stub_pool:make_nets", runner="stub_pool:run_point", record_gpu=True,
                gpu_ids=[5, 7])                      # (like --cuda_idx: the GPUs the indices stand for)
    env_before = os.environ.get("PYTHONPATH")
    os.environ["PYTHONPATH"] = HERE + os.pathsep + os.path.dirname(HERE) + (os.pathsep + env_before if env_before else "")
    try:
        log = harness.run_config(GOLD["config"], opts, workers=2, gpus=2)
    finally:
        if env_before is None:
            del os.environ["PYTHONPATH"]
        else:
            os.environ["PYTHONPATH"] = env_before
    assert list(log) == ["SetA", "SetB"]
    points = [p for ds in log.values() for seq in ds.values() for p in seq.values()]
    assert len(points) == 6 and {tuple(sorted(seq)) for ds in log.values() for seq in ds.values()} == {("000", "001")}
    for p in points:
        n = int(p["process"].rsplit("-", 1)[1])
        assert p["gpu"] == n % 2 and p["seen_visible"] == str([5, 7][n % 2]) and p["nets"] == "i_net@" + p["seen_visible"]
        assert p["qp_i"] == p["qp_p"] == (0, 63)[p["rate_idx"]]
    assert len({p["process"] for p in points}) <= 2
    buf = io.StringIO()
    harness.dump_json(log, buf)
    assert '"ave_all_frame_bpp": 0.563000' in buf.getvalue()


def test_force_intra_and_check_existing(tmp_path, monkeypatch):
    """--force_intra: every frame an I frame (intra period 1, test_video.py:490-491); --check_existing: a point whose .bin and
    .json exist with the right frame count is not coded again (test_video.py:130-137); the per-point log is written beside the
    container (test_video.py:345-346); --save_decoded_frame hands run_one_point a reconstruction path."""
    jobs = harness.jobs_from_config(GOLD["config"], dict(qp_i=[10], force_intra=True))
    assert jobs and all(j["intra_period"] == 1 for j in jobs)
    jobs = harness.jobs_from_config(GOLD["config"], dict(qp_i=[10], force_intra=True, force_intra_period=4))
    assert all(j["intra_period"] == 4 for j in jobs)               # (the explicit period wins, like in the reference)
    job = harness.jobs_from_config(GOLD["config"], dict(qp_i=[10]))[0]
    calls = []

    def fake_point(i_net, p_net, src, w, h, n, qp_i, qp_p, **kw):
        calls.append(kw)
        open(kw["bin_path"], "wb").write(b"bin")
        return {"i_frame_num": 1, "p_frame_num": n - 1, "ave_all_frame_bpp": 0.25}

    monkeypatch.setattr(harness, "run_one_point", fake_point)
    opts = dict(stream_path=str(tmp_path), check_existing=True, save_decoded_frame=True)
    first = harness.run_job(("i", "p"), job, opts)
    folder = tmp_path / job["ds_name"]
    assert (folder / f"{job['seq']}_q10.bin").exists() and json.load(open(folder / f"{job['seq']}_q10.json")) == first
    assert calls[0]["rec_path"] == str(folder / f"{job['seq']}_q10.yuv") and calls[0]["bin_path"].endswith("_q10.bin")
    again = harness.run_job(("i", "p"), job, opts)
    assert again == first and len(calls) == 1                        # served from the stored log
    harness.run_job(("i", "p"), dict(job, frame_num=job["frame_num"] + 1), opts)
    assert len(calls) == 2                                           # a log with another frame count is not trusted
    harness.run_job(("i", "p"), job, dict(opts, check_existing=False))
    assert len(calls) == 3
