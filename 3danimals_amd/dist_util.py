"""Data-parallel plumbing shared by bench.py and the tests (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" on CPU).

The hot path shards over the batch and exchanges nothing; the only collective of a training step is
the gradient all-reduce of the MLP parameters, which under the Trainer is done by DistributedDataParallel.  These helpers
cover what surrounds it: rank-local batch shards, the barrier-bracketed max-over-ranks timing the bench contract asks for,
and an explicit bucketed gradient average for callers that run the step outside a DDP wrapper.
"""
from __future__ import annotations

import time

import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous, near-equal split of a batch: rank r owns [lo, hi)."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sync(device=None):
    """barrier + device synchronise, the bracket of the timed region."""
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda"):
        torch.cuda.synchronize()


def timed_steps(step_fn, steps: int, warmup: int, device=None, per_rank: bool = False):
    """Run ``warmup`` untimed and exactly ``steps`` timed calls; return the MAX over ranks of the elapsed seconds.
    ``per_rank``: -> (that maximum, [every rank's own seconds spent issuing and finishing its steps BEFORE the closing barrier]) -- the
    spread shows which rank the others wait for (unequal covered-pixel counts, a host stall)."""
    for _ in range(warmup):
        step_fn()
    sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    own = None
    if per_rank:
        if torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda"):
            torch.cuda.synchronize()
        own = time.perf_counter() - t0
    sync(device)
    elapsed = time.perf_counter() - t0
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not per_rank:
        return elapsed
    if not multi:
        return elapsed, [own]
    ws = dist.get_world_size()
    t = torch.zeros(ws, dtype=torch.float64, device=device if device is not None else "cpu")
    t[dist.get_rank()] = own
    dist.all_reduce(t)
    return elapsed, [float(v) for v in t.tolist()]


def allreduce_mean_grads(params, bucket_bytes: int = 25 << 20):
    """Average .grad over ranks in flat buckets (what DDP does, for steps that bypass the DDP wrapper).

    Parameters whose grad is None contribute zeros, so every rank issues the same collectives.
    """
    rank, ws = world()
    if ws == 1:
        return
    params = [p for p in params if p.requires_grad]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= ws
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        bucket, size = [], 0

    for p in params:
        bucket.append(p)
        size += p.numel() * p.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
