"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI).

The hot path shards by *permutation range* (nhood, autocorr) or *row-tile range* (co-occurrence, Ripley);
every rank holds the full (small) inputs, so the only exchange is one all-reduce of exact integer
accumulators (< 1 MB, latency-bound).  torch is imported lazily and only when a process group exists, so
the single-GPU product path has no torch dependency."""

from __future__ import annotations

import os

import numpy as np


def is_distributed() -> bool:
    force = os.environ.get("SQGR_DIST_FORCE") == "1"  # exercise the collective path with a single rank (tests)
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not force:
        return False
    try:
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return False
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) when no process group is initialised."""
    if not is_distributed():
        return 0, 1
    import torch.distributed as dist

    return dist.get_rank(), dist.get_world_size()


def shard_range(n: int, rank: int, world_size: int, begin: int = 0) -> tuple[int, int]:
    """Contiguous split of [begin, begin+n) (same chunking as ``np.array_split`` / the reference's
    contiguous per-job chunks, /root/reference/src/squidpy/_utils.py:223-231)."""
    base, rem = divmod(n, world_size)
    lo = begin + rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allreduce_sum_(arrays: list[np.ndarray]) -> list[np.ndarray]:
    """Exact all-reduce(sum) of 64-bit integer arrays (uint64 is summed modulo 2**64 via its int64 view).
    No-op without a process group."""
    if not is_distributed():
        return arrays
    import torch
    import torch.distributed as dist

    backend = dist.get_backend()
    flat = np.concatenate([np.ascontiguousarray(a).view(np.int64).reshape(-1) for a in arrays])
    t = torch.from_numpy(flat.copy())
    if backend == "nccl":
        t = t.cuda(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    flat = t.cpu().numpy()
    out, off = [], 0
    for a in arrays:
        out.append(flat[off : off + a.size].view(a.dtype).reshape(a.shape).copy())
        off += a.size
    return out


def allgather_object(a: np.ndarray) -> list[np.ndarray]:
    """One copy of ``a`` from every rank, in rank order (autocorr: feature blocks are gathered, not reduced)."""
    if not is_distributed():
        return [a]
    import torch.distributed as dist

    objs: list = [None] * dist.get_world_size()
    dist.all_gather_object(objs, a)
    return objs
