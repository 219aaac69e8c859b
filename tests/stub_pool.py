"""Stub codec for the CPU test of the harness's job fan-out (tests/test_harness_pool.py): no GPU, no models - every job
returns a small deterministic result plus what the worker process saw of its GPU assignment."""
import multiprocessing
import os


def make_nets(opts):
    return ("i_net@" + os.environ.get("HIP_VISIBLE_DEVICES", "none"), "p_net")


def run_point(nets, job, opts):
    return {"frame_pixel_num": job["src_width"] * job["src_height"], "frames": job["frame_num"],
            "ave_all_frame_bpp": 0.001 * job["qp_i"] + 0.5, "src_path": job["src_path"], "intra_period": job["intra_period"],
            "seen_visible": os.environ.get("HIP_VISIBLE_DEVICES"), "nets": nets[0],
            "process": multiprocessing.current_process().name}
