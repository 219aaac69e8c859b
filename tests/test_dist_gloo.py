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


def _bench_worker(rank, world, port, q):
    """bench.py's N > 1 control flow (measure(): warm-up + GOP alignment, barrier, timed region, barrier, MAX
    reduce) on gloo / CPU with a stub in place of the codec: rank r takes (r + 1) ms per frame."""
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cpus = dist_utils.pin_rank_threads(rank, world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def run(n, timed):
        calls.append((n, timed))
        time.sleep(0.001 * (rank + 1) * n)

    steps, warmup = 20, 5
    t0 = time.perf_counter()
    elapsed, align, mine = bench.measure(run, steps, warmup, world, torch.device("cpu"), lambda: None)
    assert 0.0 < mine <= elapsed
    wall = time.perf_counter() - t0
    q.put((rank, elapsed, align, calls, sorted(cpus), wall))
    dist.destroy_process_group()


def test_bench_control_flow_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, e0, a0, c0, cpu0, _), (r1, e1, a1, c1, cpu1, _) = res
    assert e0 == e1 and e0 >= 0.002 * 20 * 0.9                     # MAX over ranks: the slower rank's 2 ms per frame
    assert a0 == a1 == (22 - 5) % 32                               # 20 timed frames start at frame 22: the I frame (32) is inside
    assert c0 == c1 == [(5 + a0, False), (20, True)]              # exactly K timed steps after warm-up + alignment
    assert cpu0 and cpu1
    if len(os.sched_getaffinity(0)) >= 2:                          # (a one-CPU mask leaves nothing to split)
        assert not (set(cpu0) & set(cpu1))                         # the two ranks were given disjoint cores


def test_rank_cpu_partition():
    allowed = list(range(64))
    # no topology: contiguous slices
    parts = [dist_utils.rank_cpus(r, 8, allowed, gpu_cpus=[], quota=None) for r in range(8)]
    assert sorted(sum(parts, [])) == allowed and all(len(p) == 8 for p in parts)
    # two NUMA nodes with four GPUs each: every rank stays on its GPU's node, ranks of a node split it
    node = [set(range(0, 32))] * 4 + [set(range(32, 64))] * 4
    parts = [dist_utils.rank_cpus(r, 8, allowed, gpu_cpus=node, quota=None) for r in range(8)]
    assert all(set(parts[r]) <= node[r] and len(parts[r]) == 8 for r in range(8))
    assert sorted(sum(parts, [])) == allowed
    # local CPUs outside the allowed set (container cpuset): fall back to slices of what is allowed
    parts = [dist_utils.rank_cpus(r, 2, list(range(8)), gpu_cpus=[set(range(100, 110))] * 2, quota=None) for r in range(2)]
    assert parts == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert dist_utils.rank_cpus(0, 1, allowed, gpu_cpus=[], quota=None) == allowed
    # a cgroup CPU quota below the mask (the GPU boxes: 256 hardware threads visible, 16 CPUs granted): a rank keeps its
    # threads on its share of the quota inside its slice
    parts = [dist_utils.rank_cpus(r, 8, list(range(256)), gpu_cpus=[], quota=16) for r in range(8)]
    # (never fewer than MIN_RANK_CPUS: the quota limits CPU time, not parallelism - ADVICE round 3)
    assert all(len(p) == dist_utils.MIN_RANK_CPUS for p in parts) and len({c for p in parts for c in p}) == 8 * dist_utils.MIN_RANK_CPUS
    parts32 = [dist_utils.rank_cpus(r, 2, list(range(256)), gpu_cpus=[], quota=64) for r in range(2)]
    assert all(len(p) == 32 for p in parts32)
    assert [p[0] for p in parts] == [32 * r for r in range(8)]
    assert dist_utils.cpus_granted(8, list(range(256)), quota=16) == 2 and dist_utils.cpus_granted(1, list(range(8)), quota=None) == 8


def test_visible_devices_remap_disables_the_sysfs_order(monkeypatch):
    """HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES remap device indices: sysfs PCI order is then not the HIP order, and
    rank_cpus falls back to plain slices instead of pinning ranks to another GPU's NUMA node."""
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3,1")
    monkeypatch.setattr(dist_utils, "gpu_local_cpus", lambda *a, **k: (_ for _ in ()).throw(AssertionError("sysfs order used")))
    assert dist_utils.rank_cpus(1, 2, list(range(8)), quota=None) == [4, 5, 6, 7]


def _run_bench(argv, extra_env=None, timeout=300):
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DCVC_BENCH_STUB="1", **(extra_env or {}))
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, [json.loads(ln) for ln in lines]


def test_bench_gpus2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` - the shape of the driver's command - with no launcher around it: bench.py starts the two
    ranks as a child torchrun (gloo + the stub codec here: DCVC_BENCH_STUB=1), rank 0 prints ONE line with n_gpus 2."""
    p, lines = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 20 and out["warmup"] == 5
    assert out["data"] == "stub" and out["weights_identical_on_all_ranks"] is True
    # rank r sleeps (r + 1) ms per frame: the job runs at the slower rank's pace, both ranks count
    assert out["rank_fps"]["min"] < out["rank_fps"]["max"]
    assert out["rank_fps"]["min"] <= 500.0 * 1.05
    assert out["value"] <= 2 * out["rank_fps"]["min"] * 1.05
    assert out["config"]["alignment_frames"] == (22 - 5) % 32 and out["config"]["frames_run"] == 5 + 17 + 20


def test_bench_gpus1_stays_one_process():
    p, lines = _run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["rccl_ranks"] == 1


def test_bench_rejects_world_size_mismatch():
    """a launcher that started a different number of ranks than --gpus says: refuse, do not print a mislabelled line"""
    p, lines = _run_bench(["--gpus", "4"], extra_env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and not lines
    assert "--gpus 4" in p.stderr
