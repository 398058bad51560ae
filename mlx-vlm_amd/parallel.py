"""Data-parallel serving over the 8 GPUs of one MI355X node.

The reference has no multi-device path for this model (Qwen2-VL defines no
shard()), and requests are independent units, so the path
shards by REQUEST: one process per GPU (torch.distributed, backend "nccl" ==
RCCL over xGMI), every rank holds a full weight replica (2B: 4.4 GB of 288 GB),
and there is NO collective in the prefill / decode step.  Collectives are used
exactly twice per job:
  * load: rank 0 materialises the checkpoint and the flat weight buckets are
    broadcast over xGMI (ring/tree chosen by RCCL; buckets of `bucket_bytes` so a
    4.4 GB replica is a handful of large transfers, not 700 small ones)
  * end: token-id lists are gathered on rank 0 (KBs, object gather on the host).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Sequence

import torch
import torch.distributed as dist


def world() -> tuple:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, ws, local = world()
    if ws > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://")
    return rank, ws, local


def shard_requests(n_requests: int, rank: int, world_size: int, lengths: Sequence[int] | None = None) -> List[int]:
    """Request i -> rank (position in the length-sorted order) mod W, the reference's own length sort
    (ar.py:2620-2623) applied before dealing so every rank sees a similar length mix."""
    order = list(range(n_requests))
    if lengths is not None:
        order.sort(key=lambda i: (lengths[i], i))
    return [order[j] for j in range(rank, n_requests, world_size)]


def broadcast_weights(weights: Dict[str, torch.Tensor], src: int = 0, bucket_bytes: int = 1 << 30) -> Dict[str, torch.Tensor]:
    """Broadcast a name->tensor dict from `src` in large flat buckets (per dtype).  Every rank must
    pass tensors of the right shape/dtype/device (contents are overwritten on non-src ranks)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return weights
    by_dtype: Dict[torch.dtype, List[str]] = {}
    for k in sorted(weights):
        by_dtype.setdefault(weights[k].dtype, []).append(k)
    for dt, names in by_dtype.items():
        esz = torch.empty((), dtype=dt).element_size()
        bucket: List[str] = []
        nbytes = 0

        def flush():
            nonlocal bucket, nbytes
            if not bucket:
                return
            flat = torch.cat([weights[k].reshape(-1) for k in bucket])
            dist.broadcast(flat, src=src)
            off = 0
            for k in bucket:
                n = weights[k].numel()
                weights[k].copy_(flat[off:off + n].view_as(weights[k]))
                off += n
            bucket, nbytes = [], 0

        for k in names:
            sz = weights[k].numel() * esz
            if bucket and nbytes + sz > bucket_bytes:
                flush()
            bucket.append(k)
            nbytes += sz
        flush()
    return weights


def gather_results(local: List, dst: int = 0) -> List[List] | None:
    """Gather per-rank python result lists on `dst` (host side, small)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, out, dst=dst)
    return out


def max_over_ranks(x: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sum_over_ranks(x: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown():
    """Tear the process group down (no-op for a single process)."""
    if dist.is_initialized():
        dist.destroy_process_group()
