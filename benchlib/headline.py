"""The timed region of the default line (its own process: bench.py --stage headline, or one rank of the N-rank job)."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch
from .common import *  # noqa: F401,F403

__all__ = ['stage_headline', '_dump_address_map']


def stage_headline(args, rank, ws, local):
    """The timed region of the default line: W warm-up passes, then exactly K passes of BASELINE configs[1] bracketed by a
    barrier + synchronize on both sides, MAX over ranks.  -> the line's dict (rank 0; the other ranks get the same numbers)."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.models import qwen2_vl

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rank == 0 and ws > 1:
        print(f"[bench] {ws} ranks, backend {_dist_info(ws)['backend']} (RCCL over xGMI), one process per GPU", file=sys.stderr, flush=True)
    t_host0 = time.perf_counter()
    cfg, model, load = _load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, rank, dev, kv_pool_tokens=32768, max_seqs=40)
    if rank == 0 and ws > 1:
        print(f"[bench] weights: {load['weight_bytes'] / 1e9:.2f} GB broadcast from rank 0 in {load['broadcast_s']:.3f} s", file=sys.stderr, flush=True)

    t_prep0 = time.perf_counter()
    req = build_request(cfg, 448, 128, seed=rank)
    req = (req[0], req[1].to(dev), req[2])
    host_prep_s = time.perf_counter() - t_prep0
    if os.environ.get("VLM_DEBUG_ADDR"):
        _dump_address_map(model, "after load")
    for _ in range(args.warmup):
        run_step(model, req, args.max_tokens, args.lookahead)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pre_s = dec_s = 0.0
    ntok = 0
    for _ in range(args.steps):
        a, b, toks = run_step(model, req, args.max_tokens, args.lookahead)
        pre_s += a
        dec_s += b
        ntok += len(toks)
    torch.cuda.synchronize()
    parallel.barrier()
    wall = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    dec_max = parallel.max_over_ranks(dec_s, dev)
    pre_max = parallel.max_over_ranks(pre_s, dev)
    prep_max = parallel.max_over_ranks(host_prep_s, dev)
    prep_ranks, dec_ranks = parallel.per_rank(host_prep_s, dev), parallel.per_rank(dec_s, dev)
    decode_steps = args.steps * (args.max_tokens - 1)          # tokens produced by decode steps, per rank
    decode_tps = ws * decode_steps / dec_max
    ms_per_step = wall / args.steps * 1e3
    us_per_token = dec_max / decode_steps * 1e6

    lm_params = 28 * 46797824 + 1536 + 233373696
    ctx_mid = int(req[0].shape[1]) + args.max_tokens // 2
    bytes_per_token = 2 * lm_params + 28672 * ctx_mid + 28672
    step_gbs = bytes_per_token / (us_per_token * 1e-6) / 1e9
    traffic_gu, traffic_tok, traffic_src = pmc_traffic()
    dist = _dist_info(ws, load)
    dist["host_prep_s_max_over_ranks"] = prep_max          # image processing + request assembly of one rank's request
    dist["host_prep_s_per_rank"] = prep_ranks              # (the first real 8-GPU run answers "is a rank's host side the tail?" from
    dist["decode_s_per_rank"] = dec_ranks                  #  this line alone; weight_broadcast_GBps sits beside them when ranks > 1)
    out = {
        "metric": "decode tokens/sec + vision-prefill images/sec, Qwen2-VL-2B", "value": decode_tps, "unit": "tokens/s",
        "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "Qwen2-VL-2B-Instruct dims (random-init bf16), batch=1 per GPU, one 448x448 image "
                               "(1024 patches -> 256 image tokens) + 128 text tokens, greedy 256-token decode, EOS disabled",
                   "prompt_tokens": int(req[0].shape[1]), "max_tokens": args.max_tokens, "parallelism": f"dp{ws}",
                   "decode_lookahead": args.lookahead, "decode_tuning": dict(model.language_model.tuning)},
        "decode_us_per_token": us_per_token,
        "prefill_ms_to_first_token": pre_max / args.steps * 1e3,
        "prompt_tps": ws * args.steps * int(req[0].shape[1]) / pre_max,
        "e2e_tokens_per_s": ws * ntok / wall,
        "load_s": load["load_s"], "load": load, "distributed": dist,
        # the number the north-star's 60 % target refers to: the WHOLE decode step against the HBM roofline
        "roofline": {"bound": "hbm", "kernel": "whole decode step (all launches of one token)", "achieved": step_gbs,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS, "traffic": traffic_tok,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_token": bytes_per_token},
    }
    out["roofline_decode_step"] = dict(out["roofline"])
    from mlx_vlm_amd import ops as _ops
    # rows whose logits held no finite value in any greedy tail of this process (0 on a healthy run; csrc/sample.hip)
    out["decode_nan_rows"] = sum(_ops.bad_argmax_rows(st.sample_ws) for st in model.language_model._decode_states.values())
    out["_traffic_gate_up"] = traffic_gu
    return out


def _dump_address_map(model, when):
    """diagnostics (VLM_DEBUG_ADDR=1): where everything lives, so that the address of a GPU memory fault can be attributed"""
    from mlx_vlm_amd import _lib

    lm = model.language_model
    rows = [(seg["address"], seg["address"] + seg["total_size"], f"torch segment ({seg['segment_type']})") for seg in torch.cuda.memory_snapshot()]
    for name, t in (("weight arena", lm.warena.buf if lm.warena is not None else None), ("small arena", lm.arena.buf),
                    ("kpool", lm.pool.kpool), ("vpool", lm.pool.vpool)):
        if t is not None:
            rows.append((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), name))
    if _lib._ring is not None:
        rows.append((_lib._ring.buf.data_ptr(), _lib._ring.buf.data_ptr() + _lib._ring.buf.numel(), "pinned upload ring (host)"))
    print(f"[bench] address map {when}:", file=sys.stderr)
    for a, b, name in sorted(rows):
        print(f"[bench]   {a:#x} .. {b:#x}  {(b - a) / 2**20:10.2f} MiB  {name}", file=sys.stderr)
    sys.stderr.flush()
