"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm).

Inference / throughput: scenes are independent -> the scene batch is SHARDED over ranks and nothing
crosses devices on the data path; results are gathered on the host only if the caller asks.
Training: exactly one exchange per step -- a bucketed sum all-reduce of the gradients (the reference relies on
Lightning DDP, src/main.py:110-115).  xGMI is point-to-point (7 links/GPU), so buckets are large (default 64 MiB:
few, big ring collectives) and parameters that received no gradient (scratch.refinenet4.resConfUnit1 in both DPT
heads) are zero-filled so every rank reduces identical buckets without find_unused_parameters.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of `n_items` for `rank` (first n%world ranks get one more)."""
    q, r = divmod(n_items, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Slice every tensor of a (nested) batch dict along dim 0 to this rank's scenes."""
    def cut(v):
        if isinstance(v, dict):
            return {k: cut(x) for k, x in v.items()}
        if torch.is_tensor(v):
            r = shard_range(v.shape[0], rank, world)
            return v[r.start:r.stop]
        return v
    return cut(batch)


def gather_scenes(t: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-rank scene results (possibly uneven) back to the full batch order; used off the hot path."""
    world = dist.get_world_size(group)
    sizes = [len(shard_range(n_total, r, world)) for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx, *t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], 0)


def bucketed_allreduce_grads(params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, average: bool = True,
                             group=None) -> int:
    """Sum (or average) gradients across ranks in flat buckets.  Returns the number of collectives issued."""
    world = dist.get_world_size(group)
    ps: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    n_coll = 0
    bucket: List[torch.nn.Parameter] = []
    size = 0

    def flush():
        nonlocal n_coll, bucket, size
        if not bucket:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        n_coll += 1
        bucket, size = [], 0

    for p in ps:
        b = p.numel() * p.element_size()
        if size and size + b > bucket_bytes:
            flush()
        bucket.append(p)
        size += b
    flush()
    return n_coll
