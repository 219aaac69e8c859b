"""The RCCL calls of bench.py's N > 1 path on real hardware, from a ONE-GPU box: a process group of one rank on the "nccl" backend
(= RCCL on ROCm) runs the same init (with a device id), the one-blob weight broadcast, the fp64 MAX all-reduce of the timed
region, the per-rank gather and the barrier that `opendcvc_amd/dist.py` issues at N = 8.  It cannot show scaling; it shows that
every collective of the path loads, initialises and completes on this image.  One child process (its process group must not
outlive the test)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, "@REPO@")
from opendcvc_amd import dist as dist_utils, weights, _lib
_lib.require_gpu()
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=device)          # bench.py main(): the same call
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist_utils.FORCE_COLLECTIVES = True
for name in ("dmci", "dmc"):
    sd = weights.make_state_dict(name, 1234)
    t0 = time.perf_counter()
    got = dist_utils.broadcast_state_dict(name, sd, device, 0, 1)
    dt = time.perf_counter() - t0
    assert got is not sd and set(got) == set(sd)
    for k in sd:
        assert np.array_equal(np.asarray(sd[k], np.float32), got[k]), k
    print("broadcast", name, "%.3f s" % dt, flush=True)
dist_utils.barrier(1)
assert dist_utils.max_over_ranks(1.25, device, 1) == 1.25
assert dist_utils.gather_over_ranks(3.5, device, 1) == [3.5]
dist_utils.barrier(1)
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL-OK", flush=True)
"""


def test_rccl_collectives_of_the_rank_path_in_a_world_of_one():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD.replace("@REPO@", REPO)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    assert "RCCL-OK" in p.stdout and p.stdout.count("broadcast") == 2
