"""The N > 1 path of bench.py with the REAL codecs, on the one GPU a test box has: `python bench.py --gpus 2` through its own
launcher with DCVC_BENCH_REHEARSE=1 (both ranks share GPU 0 and talk over gloo - RCCL refuses two ranks on one device; everything
else is the multi-GPU run: pinning, weight broadcast, measure(), per-rank gather, rank 0's line).  Two processes on the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_rehearsal_on_one_gpu():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DCVC_BENCH_REHEARSE="1")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["metric"].startswith("REHEARSAL") and out["config"]["weights_via"] == "gloo broadcast from rank 0"
    assert out["rccl_ranks"] == 0                      # (gloo, not RCCL: a rehearsal says so)
    assert 0 < out["rank_fps"]["min"] <= out["rank_fps"]["max"]
    # whole-job value = both ranks' frames over the slower rank's time
    assert out["value"] <= 2 * out["rank_fps"]["max"] * 1.02 and out["value"] >= 2 * out["rank_fps"]["min"] * 0.9
    assert "roofline" in out and "cpu_baseline" not in out and "exact_mode" not in out     # N > 1: no CPU / exact-mode legs
    assert out["gop_bpp"] > 0 and abs(out["reference_gop"]["bpp_rel_dev"]) < 1e-3          # rank 0's stream is the bench GOP
