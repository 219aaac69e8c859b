"""The N > 1 path of bench.py (weight broadcast from rank 0, MAX reduction of the timed region)
with world_size 2 on the gloo backend (CPU)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from opendcvc_amd import dist as dist_utils
from opendcvc_amd import weights


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = weights.make_state_dict("dmc", 77) if rank == 0 else None
    got = dist_utils.broadcast_state_dict("dmc", sd, torch.device("cpu"), rank, world)
    ref = weights.make_state_dict("dmc", 77)
    ok = list(got) == list(ref) and all(np.array_equal(got[k], ref[k]) for k in ref)
    dist_utils.barrier(world)
    t = dist_utils.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
    q.put((rank, ok, t))
    dist.destroy_process_group()


def test_broadcast_and_max_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.0, 2.0]


def test_single_rank_is_passthrough():
    sd = {"a": np.zeros(3, np.float32)}
    assert dist_utils.broadcast_state_dict("dmc", sd, torch.device("cpu"), 0, 1) is sd
    assert dist_utils.max_over_ranks(3.5, torch.device("cpu"), 1) == 3.5
