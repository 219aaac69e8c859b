"""Multi-GPU plumbing: the DCVC-RT path shards by stream (one process per GPU, no data-path
collective - the reference does the same with a process pool, test_video.py:381-414,472-510).
The only communication is a one-time broadcast of the weights from rank 0 (RCCL over xGMI when the
backend is "nccl"; gloo on CPU in the tests) and the MAX reduction of the timed region."""
import glob
import os

import numpy as np
import torch

from . import arch


def _parse_cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.update(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_local_cpus(sysfs="/sys/class/drm"):
    """CPUs local to each AMD GPU (NUMA affinity of its PCI function), in PCI-address order - the order in which the
    HIP runtime enumerates devices by default.  Read from sysfs: no GPU call is made.  [] if not available."""
    gpus = []
    for dev in glob.glob(os.path.join(sysfs, "renderD*", "device")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
            bdf = os.path.basename(os.path.realpath(dev))
            gpus.append((bdf, _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())))
        except (OSError, ValueError):
            continue
    return [c for _, c in sorted(gpus)]


def cgroup_cpu_quota(path="/sys/fs/cgroup/cpu.max"):
    """CPUs' worth of time the cgroup grants this process (cgroup v2 cpu.max = quota / period), or None without a limit.
    A GPU box shows the host's 256 hardware threads in the affinity mask and grants a 16-CPU share."""
    try:
        quota, period = open(path).read().split()[:2]
        if quota != "max":
            return max(1, int(round(int(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return None


def cpus_granted(local_world, allowed=None, quota="auto"):
    """CPUs one of `local_world` ranks on this node can count on: its share of the affinity mask, cut down to its share
    of the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0) if allowed is None else allowed)
    q = cgroup_cpu_quota() if quota == "auto" else quota
    if q is not None:
        n = min(n, q)
    return max(1, n // max(1, local_world))


MIN_RANK_CPUS = 4        # encoder thread, decoder thread, one rANS worker each at least


def rank_cpus(local_rank, local_world, allowed=None, gpu_cpus=None, quota="auto"):
    """The CPU set rank `local_rank` of `local_world` ranks on this node should run on: the allowed CPUs local to its
    GPU's NUMA node, divided evenly between the ranks whose GPUs share that node; without topology information (or
    when the local CPUs are not among the allowed ones) a contiguous 1/local_world slice of the allowed CPUs."""
    allowed = sorted(os.sched_getaffinity(0) if allowed is None else allowed)
    if gpu_cpus is None:
        # sysfs lists the GPUs in PCI order = the HIP device order only while no *_VISIBLE_DEVICES variable remaps it
        remapped = any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        gpu_cpus = [] if remapped else gpu_local_cpus()
    # A cgroup CPU quota below the mask (16 CPUs of time on a 256-thread mask): a rank keeps its threads on as many CPUs
    # of its slice as its share of the quota is worth - spreading them over the whole slice buys no CPU time - but never on
    # fewer than MIN_RANK_CPUS: the quota limits CPU time, not parallelism, and a rank runs two pipeline threads plus the
    # rANS workers of two coders, whose point is to overlap each other.
    keep = max(cpus_granted(local_world, allowed, quota), MIN_RANK_CPUS)
    part = None
    if len(gpu_cpus) >= local_world and all(gpu_cpus[r] & set(allowed) for r in range(local_world)):
        mine = sorted(gpu_cpus[local_rank] & set(allowed))
        peers = [r for r in range(local_world) if gpu_cpus[r] == gpu_cpus[local_rank]]
        k, n = peers.index(local_rank), len(peers)
        part = mine[k * len(mine) // n:(k + 1) * len(mine) // n]
    if not part:
        n = len(allowed)
        part = allowed[local_rank * n // local_world:(local_rank + 1) * n // local_world]
    part = part or allowed
    return part[:max(1, keep)] if len(part) > keep else part


def pin_rank_threads(local_rank, local_world):
    """Pins this process (every thread it starts later inherits the mask: the two pipeline threads, the rANS workers)
    to rank_cpus(...).  SURVEY 8(e): with 8 ranks x (2 Python threads + up to 8 coder workers) the host cores are the
    expected scaling limiter - without pinning the ranks' threads migrate across NUMA nodes.  Call before the first
    GPU call.  Returns the CPU list (also when the mask could not be set)."""
    cpus = rank_cpus(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        pass
    torch.set_num_threads(1)      # the reference does the same (src/utils/common.py:23): no intra-op pool per rank
    return cpus


# tests/test_gpu_rccl_world1.py sets this to run the collectives below in a process group of ONE rank: the only way to put the
# RCCL calls of the N > 1 path (init with a device id, fp32 broadcast, fp64 MAX all-reduce, all-gather, barrier) on real
# hardware from a one-GPU box
FORCE_COLLECTIVES = False


def _alone(world):
    return world == 1 and not FORCE_COLLECTIVES


def broadcast_state_dict(model_name, sd, device, rank, world):
    """sd: {name: float32 ndarray} on rank 0 (ignored elsewhere).  Returns the same dict on every
    rank after ONE broadcast of a flat fp32 blob (20.7 M / 45.7 M parameters)."""
    if _alone(world):
        return sd
    import torch.distributed as dist
    spec = arch.spec_for(model_name).items
    total = sum(int(np.prod(s)) for _, s, _ in spec)
    blob = torch.empty(total, dtype=torch.float32, device=device)
    if rank == 0:
        blob.copy_(torch.from_numpy(np.concatenate([np.asarray(sd[k], np.float32).reshape(-1) for k, _, _ in spec])))
    dist.broadcast(blob, 0)
    flat = blob.cpu().numpy()
    out, off = {}, 0
    for k, s, _ in spec:
        n = int(np.prod(s))
        out[k] = flat[off:off + n].reshape(s).copy()
        off += n
    return out


def max_over_ranks(value, device, world):
    if _alone(world):
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device, world):
    """every rank's `value` (a float), as a list in rank order, on every rank"""
    if _alone(world):
        return [float(value)]
    import torch.distributed as dist
    mine = torch.tensor([value], device=device, dtype=torch.float64)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def barrier(world):
    if not _alone(world):
        import torch.distributed as dist
        dist.barrier()
