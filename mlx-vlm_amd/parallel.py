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
    if ws > 1:
        from . import utils
        utils.fit_host_threads()          # the ranks of a node share one CPU quota: each sizes its pool for its share
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


class WeightArena:
    """ONE flat allocation per dtype that holds every tensor of a replica as a 256-byte aligned view (SURVEY section 8e:
    "one ncclBroadcast of the flat weight arena at load").  Broadcasting it is zero-copy on both sides - the collective
    reads and writes the tensors in place, in slices of `bucket_bytes` of the flat buffer (no torch.cat into a transient
    bucket, no copy back out), so a 4.4 GB replica is five 1 GiB transfers over xGMI and the receiving ranks hold exactly
    one copy of the weights at any time."""

    ALIGN = 256

    def __init__(self, meta, device):
        """meta: iterable of (name, shape, dtype) - identical (content and order) on every rank"""
        self.device = torch.device(device)
        self.flat: Dict[torch.dtype, torch.Tensor] = {}
        self._views: Dict[str, torch.Tensor] = {}
        by_dtype: Dict[torch.dtype, list] = {}
        for name, shape, dt in meta:
            dt = getattr(torch, dt) if isinstance(dt, str) else dt
            by_dtype.setdefault(dt, []).append((name, tuple(int(x) for x in shape)))
        for dt, items in by_dtype.items():
            esz = torch.empty((), dtype=dt).element_size()
            pad = max(1, self.ALIGN // esz)
            offs, n = [], 0
            for _, shape in items:
                offs.append(n)
                cnt = 1
                for x in shape:
                    cnt *= x
                n += (cnt + pad - 1) // pad * pad
            flat = torch.empty(max(n, 1), dtype=dt, device=self.device)
            self.flat[dt] = flat
            for (name, shape), off in zip(items, offs):
                cnt = 1
                for x in shape:
                    cnt *= x
                self._views[name] = flat[off:off + cnt].view(shape)

    @classmethod
    def like(cls, weights: Dict[str, torch.Tensor], device=None) -> "WeightArena":
        meta = [(k, tuple(v.shape), v.dtype) for k, v in sorted(weights.items())]
        dev = device if device is not None else next(iter(weights.values())).device
        return cls(meta, dev)

    @staticmethod
    def meta_of(weights: Dict[str, torch.Tensor]):
        """what the receiving side needs before it can allocate (picklable)"""
        return [(k, tuple(v.shape), str(v.dtype).split(".")[-1]) for k, v in sorted(weights.items())]

    def tensors(self) -> Dict[str, torch.Tensor]:
        return dict(self._views)

    def load(self, weights: Dict[str, torch.Tensor]):
        """source rank: the tensors move into their views (for a checkpoint on the host this IS the host-to-device copy)"""
        for k, v in self._views.items():
            v.copy_(weights[k], non_blocking=True)
        return self

    @property
    def nbytes(self) -> int:
        return int(sum(v.numel() * v.element_size() for v in self._views.values()))

    def broadcast(self, src: int = 0, bucket_bytes: int = 1 << 30) -> int:
        """in place; -> number of collectives issued"""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return 0
        calls = 0
        for dt in sorted(self.flat, key=str):
            flat = self.flat[dt]
            step = max(1, int(bucket_bytes) // flat.element_size())
            for off in range(0, flat.numel(), step):
                dist.broadcast(flat[off:off + step], src=src)
                calls += 1
        return calls


def broadcast_weights(weights: Dict[str, torch.Tensor], src: int = 0, bucket_bytes: int = 1 << 30) -> Dict[str, torch.Tensor]:
    """Broadcast a name -> tensor dict from `src` through a WeightArena: the dict's entries are REPLACED by views of one flat
    buffer per dtype (every rank must pass tensors of the right shape / dtype / device; their contents matter on `src`
    only), which is then broadcast in place in `bucket_bytes` slices.  One device-to-device copy of the replica on entry
    (callers that can allocate inside the arena from the start - dp_load - pay none)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return weights
    arena = WeightArena.like(weights)
    if dist.get_rank() == src:
        arena.load(weights)
    weights.update(arena.tensors())          # the original tensors are released here
    arena.broadcast(src=src, bucket_bytes=bucket_bytes)
    return weights


def dp_load(model_path: str, device=None, src: int = 0, bucket_bytes: int = 1 << 30, **kwargs):
    """`load()` for a data-parallel job (reference utils.py:1065-1119 on every rank would read the checkpoint W times):
    rank `src` reads + sanitizes the safetensors, every other rank receives the replica over RCCL/xGMI in flat buckets
    (`broadcast_weights`) and packs it for its own GPU.  -> (model, processor, stats) with stats = {"ranks", "backend",
    "weight_bytes", "broadcast_s"} so that the first multi-GPU run explains itself."""
    import time

    from . import utils

    rank, ws, local = world()
    utils.fit_host_threads()          # this rank's share of the node's CPU quota (quota // LOCAL_WORLD_SIZE), before any CPU tensor op
    if device is None:
        device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    config = utils.load_config(model_path)
    arch, _ = utils.get_model_and_args(config)
    mc = arch.ModelConfig.from_dict(config)
    model = arch.Model(mc, device=device, **kwargs)
    weights = utils.read_sanitized_weights(model_path, model, config) if rank == src else None
    stats = {"ranks": ws, "backend": dist.get_backend() if dist.is_initialized() else None, "weight_bytes": 0,
             "broadcast_s": 0.0}
    if ws > 1 and dist.is_initialized():
        # the receiving side needs names / shapes / dtypes before it can post the bucket receives
        meta = [WeightArena.meta_of(weights)] if rank == src else [None]
        dist.broadcast_object_list(meta, src=src)
        comm_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
        # every rank allocates the arena; the source fills it straight from the checkpoint tensors (its host-to-device copy),
        # the others receive in place: no staging bucket, no second copy of the replica anywhere
        arena = WeightArena(meta[0], comm_dev)
        if rank == src:
            arena.load(weights)
        weights = arena.tensors()
        if comm_dev.type == "cuda":
            torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        stats["broadcast_calls"] = arena.broadcast(src=src, bucket_bytes=bucket_bytes)
        if comm_dev.type == "cuda":
            torch.cuda.synchronize()
        stats["broadcast_s"] = max_over_ranks(time.perf_counter() - t0, comm_dev)
    stats["weight_bytes"] = int(sum(v.numel() * v.element_size() for v in weights.values()))
    model.load_weights(weights)
    del weights
    processor = utils.load_processor(model_path, model.config)
    utils.freeze_heap()
    if rank == 0 and ws > 1:
        gbs = stats["weight_bytes"] / max(stats["broadcast_s"], 1e-9) / 1e9
        print(f"[dp_load] {ws} ranks over {stats['backend']}: {stats['weight_bytes'] / 1e9:.2f} GB replica broadcast in "
              f"{stats['broadcast_s']:.3f} s ({gbs:.1f} GB/s per receiving rank)", flush=True)
    return model, processor, stats


_REQUEST_KEYS = ("input_ids", "pixel_values", "image_grid_thw", "max_tokens")     # the rest of a request: model-specific
                                                                                  # get_input_embeddings arguments (phi3_v: image_sizes)


def dp_batch_generate(model, processor, prompts=None, images=None, *, requests=None, max_tokens=128, serve=None,
                      dst: int = 0, **kwargs):
    """Data-parallel `batch_generate` (reference ar.py:2890-3096 runs one device): every rank calls this with the SAME
    request list; rank r serves the requests `shard_requests` deals it - the reference's own length sort
    (ar.py:2620-2623) first, so the ranks see similar length mixes - on its own continuous `BatchGenerator`, with no
    collective inside the prefill / decode steps; the token lists are gathered on `dst` at the end.

    requests: pre-tokenised dicts {"input_ids", "pixel_values"?, "image_grid_thw"?, "max_tokens"?} (bypass, as
    dispatch.py:759-762) - or prompts (+ one image each): every rank tokenises / image-processes ONLY the requests it is
    dealt (the deal then sorts by a tokeniser-free length proxy).
    serve(indices, requests, max_tokens) -> list of token lists: the per-rank engine (default: the continuous generator);
    replaced by a mock in the CPU tests.
    -> on `dst`: {"tokens": per request in the ORIGINAL order, "texts", "ranks", "per_rank_requests", "generation_tokens",
    "wall_s", "tokens_per_s"}; on the other ranks: None."""
    import time

    import numpy as np

    rank, ws, _ = world()
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        rank, ws = 0, 1
    mt_default = max_tokens
    if requests is None:
        # prompts: the deal needs a length BEFORE anything is tokenised - the host-side tokeniser + image processor is the
        # resource that scales worst with the rank count, so every rank prepares ONLY its own requests.
        # Proxy of the reference's token-length sort key: characters of the prompt, an image counts as 1024 characters
        # (identical on every rank, no tokeniser involved).
        from .utils import prepare_inputs

        prompts = list(prompts or [])
        images = list(images) if images is not None else [None] * len(prompts)
        n = len(prompts)
        lengths = [len(p) + (1024 if im is not None else 0) for p, im in zip(prompts, images)]
        mine = shard_requests(n, rank, ws, lengths)
        requests = [None] * n
        t_prep = time.perf_counter()
        for i in mine:
            inp = prepare_inputs(processor, images=images[i], prompts=prompts[i])
            requests[i] = {"input_ids": np.asarray(inp["input_ids"]).reshape(-1), "pixel_values": inp.get("pixel_values"),
                           "image_grid_thw": inp.get("image_grid_thw"),
                           **{k: v for k, v in inp.items() if k not in _REQUEST_KEYS and k != "attention_mask"}}
        mt = [int(mt_default)] * n
        host_prep_s = time.perf_counter() - t_prep
    else:
        host_prep_s = 0.0
        n = len(requests)
        lengths = [int(np.asarray(r["input_ids"]).size) for r in requests]
        mine = shard_requests(n, rank, ws, lengths)
        mt = [int(r.get("max_tokens", max_tokens)) for r in requests]
    if serve is None:
        def serve(indices, reqs, max_toks):
            from .batch import generate_batch_continuous

            if not indices:
                return []
            tok = getattr(processor, "tokenizer", processor)
            stop = getattr(getattr(tok, "stopping_criteria", None), "eos_token_ids", ()) or ()
            toks, st = generate_batch_continuous(model, [reqs[i]["input_ids"] for i in indices],
                                                 [reqs[i].get("pixel_values") for i in indices],
                                                 [reqs[i].get("image_grid_thw") for i in indices],
                                                 max_tokens=[max_toks[i] for i in indices], stop_ids=tuple(stop),
                                                 extras=[{k: v for k, v in reqs[i].items() if k not in _REQUEST_KEYS} for i in indices],
                                                 **kwargs)
            return toks, st
    barrier()
    t0 = time.perf_counter()
    local = serve(mine, requests, mt)
    st = None
    if isinstance(local, tuple):            # the default engine also hands back its BatchStats (a mock may return tokens only)
        local, st = local
    if torch.cuda.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        torch.cuda.synchronize()
    wall = max_over_ranks(time.perf_counter() - t0)
    # decode / prefill split of the job: tokens summed, times maximised over the ranks (what a whole-job rate divides by)
    gen_time = max_over_ranks(float(getattr(st, "generation_time", 0.0) or 0.0))
    pre_time = max_over_ranks(float(getattr(st, "prompt_time", 0.0) or 0.0))
    gen_tok = sum_over_ranks(float(getattr(st, "generation_tokens", 0) or 0))
    pre_tok = sum_over_ranks(float(getattr(st, "prompt_tokens", 0) or 0))
    dec_steps = sum_over_ranks(float(getattr(st, "decode_steps", 0) or 0))
    parts = gather_results(list(zip(mine, [list(map(int, t)) for t in local])), dst=dst)
    # per-rank host seconds spent tokenising + image-processing (W ranks share the node's host cores: this is the part of
    # the job that does not scale with the GPU count) and per-rank serving seconds
    host = gather_results([float(host_prep_s), float(time.perf_counter() - t0)], dst=dst)
    if rank != dst:
        return None
    tokens: List = [None] * n
    for part in parts:
        for i, t in part:
            tokens[i] = t
    tok = getattr(processor, "tokenizer", processor) if processor is not None else None
    texts = [tok.decode(t) if (tok is not None and hasattr(tok, "decode")) else "" for t in tokens]
    total = sum(len(t) for t in tokens)
    return {"tokens": tokens, "texts": texts, "ranks": ws, "per_rank_requests": [len(p) for p in parts],
            "generation_tokens": total, "wall_s": wall, "tokens_per_s": total / max(wall, 1e-9),
            "decode_tokens": int(gen_tok), "decode_time_s": gen_time, "prompt_tokens": int(pre_tok), "prompt_time_s": pre_time,
            "decode_steps": int(dec_steps), "host_prep_s_per_rank": [h[0] for h in host],
            "serve_s_per_rank": [h[1] for h in host]}


def gather_results(local: List, dst: int = 0) -> List[List] | None:
    """Gather per-rank python result lists on `dst` (host side, small)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, out, dst=dst)
    return out


def _scalar_dtype():
    """dtype of the scalar all-reduces below: fp32 over RCCL (the dtype every NCCL-family reduction kernel is built and
    exercised for; timings of seconds and counts below 2^24 are exact enough in it), fp64 on the gloo CPU path."""
    return torch.float32 if dist.get_backend() == "nccl" else torch.float64


def max_over_ranks(x: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=_scalar_dtype(), device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def per_rank(x: float, device=None) -> list:
    """every rank's value of `x`, in rank order, on every rank - one all-reduce (SUM) of a vector in which a rank fills only its own
    slot: the same collective as max_over_ranks / sum_over_ranks (no object gather on a path that has not seen hardware)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(x)]
    t = torch.zeros(dist.get_world_size(), dtype=_scalar_dtype(), device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    t[dist.get_rank()] = float(x)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def sum_over_ranks(x: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=_scalar_dtype(), device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def shutdown():
    """Tear the process group down (no-op for a single process)."""
    if dist.is_initialized():
        dist.destroy_process_group()
