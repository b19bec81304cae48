"""
Data-parallel helpers (new: the reference is single-process).

Training is plain data parallelism over windows: every rank holds a full replica (5.9 M fp32 parameters), computes
its own loss/gradients and the gradients are averaged with RCCL all-reduces over flat buckets after the backward
pass.  The model object is driven through `net(batch)` / `net.backward(batch, out)` like the reference's training
loop (scripts/train.py:146-150) rather than through a wrapped `forward`, so the averaging is an explicit call instead
of DistributedDataParallel hooks.  BatchNorm statistics stay per-rank, as in the reference (there is no SyncBN).
"""
import torch


def init_from_env(device=None, backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (127.0.0.1 by default). Returns (rank, world)."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if (device is not None and device.type == 'cuda') else 'gloo'
        kwargs = {'device_id': device} if backend == 'nccl' else {}
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def allreduce_gradients(parameters, bucket_bytes=8 << 20, group=None):
    """
    Average `.grad` of the given parameters over all ranks, in flat buckets of about `bucket_bytes` (a few large
    collectives instead of one per tensor: ring all-reduce over xGMI is per-link bound, small messages waste it).
    Parameters without a gradient contribute zeros so that every rank issues the same collectives.
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    params = [p for p in parameters if p.requires_grad]
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        cur.append(p)
        cur_bytes += p.numel() * p.element_size()
        if cur_bytes >= bucket_bytes:
            buckets.append(cur)
            cur, cur_bytes = [], 0
    if cur:
        buckets.append(cur)
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
    return len(buckets)


def shard_range(n_items, rank, world):
    """Contiguous, balanced [start, end) of `n_items` for `rank` (windows of a batch, BASELINE configs 2/3)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)
