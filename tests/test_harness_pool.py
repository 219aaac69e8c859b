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


REFERENCE_README_COMMAND = ("--model_path_i ./checkpoints/cvpr2025_image.pth.tar --model_path_p ./checkpoints/cvpr2025_video.pth.tar "
                            "--rate_num 4 --test_config ./dataset_config_example_yuv420.json --cuda 1 -w 1 --write_stream 1 "
                            "--force_zero_thres 0.12 --output_path output.json --force_intra_period -1 --reset_interval 64 "
                            "--force_frame_num -1 --check_existing 0 --verbose 0")


def test_the_references_command_line_is_accepted_verbatim(monkeypatch):
    """The argument list of the test command in the reference's README (README.md:166, parse_args test_video.py:30-56) parses
    under the reference's own spellings and value conventions, and means the same: manifest, four rate points, one worker,
    containers written (to the reference's default folder out_bin), reset interval 64, nothing re-used."""
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(harness, "count_gpus", lambda: 8)
    ap = harness.build_parser()
    a = ap.parse_args(REFERENCE_README_COMMAND.split())
    opts, gpus = harness.manifest_options(a, ap)
    assert a.test_config == "./dataset_config_example_yuv420.json" and a.worker == 1 and a.output_path == "output.json"
    assert gpus == 8 and opts["gpu_ids"] is None
    assert opts["model_i"].endswith("cvpr2025_image.pth.tar") and opts["model_p"].endswith("cvpr2025_video.pth.tar")
    assert opts["rate_num"] == 4 and opts["reset_interval"] == 64 and opts["force_intra_period"] == -1 and opts["force_frame_num"] == -1
    assert opts["stream_path"] == "out_bin" and opts["check_existing"] is False and opts["force_zero_thres"] == 0.12
    assert opts["verbose"] == 0 and opts["calc_ssim"] is False and opts["force_intra"] is False
    # every other option of the reference's parser, its spelling and its `--flag <bool>` convention
    a = ap.parse_args("--test_config m.json --output_path o.json --cuda True --cuda_idx 4 5 -w 4 --qp_i 10 20 --qp_p 12 22 "
                      "--force_intra true --calc_ssim 1 --save_decoded_frame yes --check_existing t --verbose_json True "
                      "--stream_path bins --force_root_path /data --write_stream False".split())
    opts, gpus = harness.manifest_options(a, ap)
    assert opts["gpu_ids"] == ["4", "5"] and gpus == 2 and a.worker == 4
    assert opts["qp_i"] == [10, 20] and opts["qp_p"] == [12, 22] and opts["force_root_path"] == "/data" and opts["stream_path"] == "bins"
    assert all(opts[k] is True for k in ("force_intra", "calc_ssim", "save_decoded_frame", "check_existing", "verbose_json"))
    # this repo's own spellings mean the same, flags without a value included
    b = ap.parse_args("--test-config m.json --output-path o.json --gpu-ids 4,5 -w 4 --qp-i 10 20 --qp-p 12 22 --force-intra --calc-ssim "
                      "--save-decoded-frame --check-existing --verbose-json --stream-path bins --force-root-path /data".split())
    assert harness.manifest_options(b, ap) == (opts, gpus)
    # no CPU path: the reference's --cuda 0 is refused, not silently run somewhere else
    with pytest.raises(SystemExit):
        harness.manifest_options(ap.parse_args("--test_config m.json --cuda 0".split()), ap)
    with pytest.raises(SystemExit):
        ap.parse_args("--test_config m.json --calc_ssim maybe".split())
