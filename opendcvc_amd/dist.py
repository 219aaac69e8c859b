"""Multi-GPU plumbing: the DCVC-RT path shards by stream (one process per GPU, no data-path
collective - the reference does the same with a process pool, test_video.py:381-414,472-510).
The only communication is a one-time broadcast of the weights from rank 0 (RCCL over xGMI when the
backend is "nccl"; gloo on CPU in the tests) and the MAX reduction of the timed region."""
import numpy as np
import torch

from . import arch


def broadcast_state_dict(model_name, sd, device, rank, world):
    """sd: {name: float32 ndarray} on rank 0 (ignored elsewhere).  Returns the same dict on every
    rank after ONE broadcast of a flat fp32 blob (20.7 M / 45.7 M parameters)."""
    if world == 1:
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
    if world == 1:
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([value], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
