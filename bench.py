#!/usr/bin/env python
"""Headline benchmark: decode tokens/sec + vision-prefill images/sec, Qwen2-VL-2B bf16 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one request of BASELINE.json configs[1]:
one synthetic 448x448 image (1024 patches -> 256 image tokens) + 128 text tokens,
ViT prefill -> projector -> LLM prefill -> 256 greedy tokens (EOS disabled), batch 1
per GPU.  N > 1: one process per GPU (RCCL), weights broadcast from rank 0 at load,
every rank serves its own request stream (weak scaling, no collective in the step).
Rank 0 prints ONE JSON line; `value` = whole-job decode tokens/s.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak
VIT_TFLOP_448 = 1.481        # SURVEY.md §8d per 448^2 image
VIT_TFLOP_336 = 0.791


def build_request(cfg, image_hw, n_text, seed):
    from mlx_vlm_amd.models.qwen2_vl.processing_qwen2_vl import Qwen2VLImageProcessor

    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (3, image_hw, image_hw), dtype=np.uint8)
    out = Qwen2VLImageProcessor()([img])
    pix, thw = out["pixel_values"], out["image_grid_thw"]
    n_img = int(thw.prod()) // 4
    text = np.random.default_rng(1000 + seed).integers(0, 151643, n_text)
    ids = np.concatenate([[cfg.vision_start_token_id], np.full(n_img, cfg.image_token_id), [cfg.vision_start_token_id + 1], text])
    return ids.astype(np.int64)[None], torch.from_numpy(pix), thw


def run_step(model, req, max_tokens, lookahead):
    """-> (seconds to first token, seconds for the remaining tokens, tokens)"""
    from mlx_vlm_amd.generate import generate_step

    ids, pix, thw = req
    t0 = time.perf_counter()
    gen = generate_step(ids, model, pix, None, max_tokens=max_tokens, temperature=0.0, image_grid_thw=thw,
                        return_logprobs=False, lookahead=lookahead)
    toks, t_first = [], None
    for tok, _ in gen:
        if t_first is None:
            t_first = time.perf_counter()
        toks.append(tok)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return t_first - t0, t1 - t_first, toks


def time_events(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


def kernel_rooflines(model, cfg):
    """HIP-event timing of the dominant decode kernels on the live weights (all 28 layers cycled, 1.5 GB > the
    256 MB Infinity Cache, so every launch streams from HBM)."""
    from mlx_vlm_amd import ops

    lm = model.language_model
    t = cfg.text_config
    D, I, V = t.hidden_size, t.intermediate_size, t.vocab_size
    x = torch.randn(1, D, device="cuda").to(torch.bfloat16)
    act = torch.randn(1, I, device="cuda").to(torch.bfloat16)
    out_gu = torch.empty(1, I, dtype=torch.bfloat16, device="cuda")
    h = torch.zeros(1, D, dtype=torch.bfloat16, device="cuda")
    logits = torch.empty(1, V, dtype=torch.bfloat16, device="cuda")
    L = t.num_hidden_layers

    def gu():
        for i in range(L):
            ops.gemv(x, lm._w[f"{i}.wgu"], norm_w=lm._w[f"{i}.ln2"], out=out_gu, epilogue=ops.EPI_SWIGLU)

    def down():
        for i in range(L):
            ops.gemv(act, lm._w[f"{i}.wdown"], res=h, out=h, epilogue=ops.EPI_RESIDUAL)

    def head():
        ops.gemv(x, lm._w["head"], norm_w=lm._w["norm"], out=logits)

    res = {}
    for name, fn, nbytes, per in (("gemv_gate_up_swiglu", gu, 2 * 2 * I * D, L), ("gemv_down_residual", down, 2 * I * D, L),
                                  ("gemv_lm_head", head, 2 * V * D, 1)):
        fn()
        torch.cuda.synchronize()
        dt = time_events(fn, 4) / per
        res[name] = {"bytes_per_launch": nbytes, "us_per_launch": dt * 1e6, "GBps": nbytes / dt / 1e9}
    return res


def vit_throughput(model, cfg, n_images, hw, reps=3):
    reqs = [build_request(cfg, hw, 1, 100 + i) for i in range(n_images)]
    pix = torch.cat([r[1] for r in reqs], dim=0).cuda()
    thw = np.concatenate([r[2] for r in reqs], axis=0)
    for _ in range(2):      # first call builds the rope tables, second settles clocks / caches
        model.vision_tower(pix, thw)
    torch.cuda.synchronize()
    dts = sorted(time_events(lambda: model.vision_tower(pix, thw), 1) for _ in range(max(reps, 5)))
    dt = dts[len(dts) // 2]   # median of single-call timings
    return n_images / dt, dt


def batch_decode_throughput(model, cfg, B=8, max_tokens=64):
    """Extra (not the headline): B concurrent requests per GPU through batch_generate_ids (one ViT call, one varlen
    prefill, batched graph decode - the weights are streamed once per step for all B rows)."""
    from mlx_vlm_amd.generate import batch_generate_ids

    reqs = [build_request(cfg, 336, 128, 500 + i) for i in range(B)]
    ids = [r[0].reshape(-1) for r in reqs]
    pix = [r[1] for r in reqs]
    thw = [r[2] for r in reqs]
    batch_generate_ids(model, ids, pix, thw, max_tokens=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, stats = batch_generate_ids(model, ids, pix, thw, max_tokens=max_tokens)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"batch": B, "image": "336x336", "max_tokens": max_tokens, "generation_tps": stats.generation_tps,
            "prompt_tps": stats.prompt_tps, "e2e_tokens_per_s": sum(len(t) for t in toks) / dt}


def continuous_batch_throughput(model, cfg, n_requests=24, rows=8):
    """Extra: a queue of requests with different lengths (336x336 image + 64-token prompt, 24..96 new tokens) through the
    continuous `BatchGenerator` (8 decode rows; rows are refilled from the queue as requests finish) vs the same queue
    as static batches of 8 that wait for their longest member."""
    from mlx_vlm_amd.batch import generate_batch_continuous
    from mlx_vlm_amd.generate import batch_generate_ids

    reqs = [build_request(cfg, 336, 64, 700 + i) for i in range(n_requests)]
    ids = [r[0].reshape(-1) for r in reqs]
    pix = [r[1] for r in reqs]
    thw = [r[2] for r in reqs]
    lens = [24 + (37 * i) % 73 for i in range(n_requests)]
    out = {"requests": n_requests, "rows": rows, "image": "336x336", "new_tokens": f"{min(lens)}..{max(lens)}"}

    def run_continuous():
        from mlx_vlm_amd.batch import BatchGenerator
        gen = BatchGenerator(model, None, completion_batch_size=rows, prefill_batch_size=rows, compute_logprobs=False)
        kw = [dict(pixel_values=p, image_grid_thw=g) for p, g in zip(pix, thw)]
        gen.insert(ids, lens, prompt_kwargs=kw)
        n = 0
        while gen.has_work:
            n += len(gen.next()[1])
        gen.close()
        return n

    def run_static():
        n = 0
        for i in range(0, n_requests, rows):     # a static batch runs to its longest member
            sl = slice(i, i + rows)
            toks, _ = batch_generate_ids(model, ids[sl], pix[sl], thw[sl], max_tokens=max(lens[sl]))
            n += sum(min(len(t), m) for t, m in zip(toks, lens[sl]))
        return n

    for name, fn in (("continuous", run_continuous), ("static", run_static)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = fn()
        torch.cuda.synchronize()
        out[name + "_useful_tokens_per_s"] = n / (time.perf_counter() - t0)
    return out


def cpu_baseline(threads):
    """Reference-equivalent CPU path (the oracle: torch-CPU restatement of the reference; the reference itself needs
    `mlx`, which is not installable here).  Bounded sample, see the returned `sample` string."""
    from oracle import ops as O
    from oracle import qwen2_vl as oq

    torch.set_num_threads(threads)
    LS, VS = 2, 2   # layers / blocks actually timed
    cfg = oq.Cfg(text=oq.TextCfg(num_hidden_layers=LS), vision=oq.VisionCfg(depth=VS))
    W = oq.random_weights(cfg, seed=0, dtype=torch.float32)
    t = cfg.text
    # --- decode: per-layer time and lm_head time for ONE token at context 384
    ctx = 384
    cache = [O.KVCache() for _ in range(LS)]
    emb = torch.randn(1, ctx, t.hidden_size) * 0.02
    pos = torch.arange(ctx)[None, None].expand(3, 1, ctx)
    oq.qwen2_model(W, cfg, emb, cache, pos)
    e1 = torch.randn(1, 1, t.hidden_size) * 0.02
    n_tok = 6
    t0 = time.perf_counter()
    for i in range(n_tok):
        h = oq.qwen2_model(W, cfg, e1, cache, torch.full((3, 1, 1), ctx + i))
    t_layers = (time.perf_counter() - t0) / n_tok
    t0 = time.perf_counter()
    for i in range(3):
        oq.lm_head(W, cfg, h)
    t_head = (time.perf_counter() - t0) / 3
    tok_s = 1.0 / (t_layers / LS * 28 + t_head)
    # --- ViT: per-block time on one 448^2 image (1024 patches)
    grid = np.array([[1, 32, 32]])
    pix = torch.randn(1024, 1176)
    x = oq.patch_embed(W, cfg, pix)
    freqs = O.vision_rotary_freqs(grid, 80)
    cu = oq.vision_cu_seqlens(grid)
    t0 = time.perf_counter()
    for i in range(VS):
        x = oq.vision_block(W, i, cfg, x, cu, freqs)
    t_block = (time.perf_counter() - t0) / VS
    t0 = time.perf_counter()
    oq.patch_merger(W, cfg, x)
    t_merge = time.perf_counter() - t0
    img_s = 1.0 / (t_block * 32 + t_merge)
    return {"value": tok_s, "unit": "tokens/s", "cores": threads, "kind": "port",
            "vision_images_per_s": img_s,
            "sample": (f"oracle (torch-CPU fp32 restatement of the reference), Qwen2-VL-2B dims, {threads} threads: decode = "
                       f"{n_tok} tokens through {LS} of 28 LLM layers at ctx {ctx} + lm_head, per-token time extrapolated "
                       f"x28/{LS} layers; vision = {VS} of 32 ViT blocks + merger on one 448x448 image, extrapolated x32/{VS}")}


DECODE_KERNELS = ("gemv_rowwave_kernel", "gemv_splitk_kernel", "attn_decode_mfma_kernel", "attn_decode_combine_kernel",
                  "lse_partial_kernel", "logprob_argmax_kernel", "argmax_final_kernel", "embed_gather_kernel",
                  "decode_advance_kernel", "sample_filter_kernel")
GATE_UP_KERNEL = "gemv_rowwave_kernel<4, 3, 1, 1, 16>"     # name as rocprofv3 prints it (R=4 rows/wave, RMSNorm prologue, SwiGLU)


def pmc_traffic():
    """HBM bytes per launch from the committed --pmc pass (scripts/final_round.sh -> profiles/r01_pmc_traffic.json:
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of MI355X_MICROARCH.md).  bench.py cannot collect
    hardware counters itself; None when the file is absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    gu = d.get(GATE_UP_KERNEL, {}).get("hbm_bytes_per_launch")
    steps = d.get("decode_advance_kernel", {}).get("launches", 0)
    per_tok = None
    if steps:
        per_tok = sum(v.get("hbm_bytes_per_launch", 0.0) * v["launches"] for k, v in d.items()
                      if k.startswith(DECODE_KERNELS)) / steps
    return gu, per_tok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--max-tokens", type=int, default=256)
    ap.add_argument("--lookahead", type=int, default=8)
    ap.add_argument("--vit-batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip kernel rooflines / ViT throughput (profiling runs)")
    args = ap.parse_args()

    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.models.qwen2_vl import Model, ModelConfig

    rank, ws, local = parallel.init()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))

    # rank 0 materialises the replica; the others receive it over RCCL/xGMI
    t0 = time.perf_counter()
    W = synthetic.random_weights(cfg, seed=0, device=dev, fill=(rank == 0))
    parallel.broadcast_weights(W, src=0)
    model = Model(cfg, device=dev, kv_pool_tokens=16384, max_seqs=16)
    model.load_weights(W)
    del W
    torch.cuda.synchronize()
    load_s = time.perf_counter() - t0
    from mlx_vlm_amd.utils import freeze_heap
    freeze_heap()          # what load() does: no 100 ms cyclic-GC passes over the import heap inside timed loops

    req = build_request(cfg, 448, 128, seed=rank)
    req = (req[0], req[1].to(dev), req[2])
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
    decode_steps = args.steps * (args.max_tokens - 1)          # tokens produced by decode steps, per rank
    decode_tps = ws * decode_steps / dec_max
    ms_per_step = wall / args.steps * 1e3
    us_per_token = dec_max / decode_steps * 1e6

    extras = {}
    if rank == 0 and not args.no_extras:
        kr = kernel_rooflines(model, cfg)
        ips336, dt336 = vit_throughput(model, cfg, args.vit_batch, 336)
        ips448, dt448 = vit_throughput(model, cfg, 1, 448)
        extras = dict(kernels=kr, vit336=(ips336, dt336), vit448=(ips448, dt448))
        # exploratory single-GPU extras: not part of the scaling runs (the other ranks would only wait for rank 0)
        for key, fn in (("batch8", lambda: batch_decode_throughput(model, cfg, 8, 64)),
                        ("continuous", lambda: continuous_batch_throughput(model, cfg))):
            if ws > 1:
                extras[key] = None
                continue
            try:
                extras[key] = fn()
            except Exception as e:   # an extra must never cost the headline line
                extras[key] = {"error": f"{type(e).__name__}: {e}"}
    cpu = None
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(min(os.cpu_count() or 1, 32))

    if rank == 0:
        t = cfg.text_config
        lm_params = 28 * 46797824 + 1536 + 233373696
        ctx_mid = int(req[0].shape[1]) + args.max_tokens // 2
        bytes_per_token = 2 * lm_params + 28672 * ctx_mid + 28672
        step_gbs = bytes_per_token / (us_per_token * 1e-6) / 1e9
        traffic_gu, traffic_tok = pmc_traffic()
        out = {
            "metric": "decode tokens/sec + vision-prefill images/sec, Qwen2-VL-2B", "value": decode_tps, "unit": "tokens/s",
            "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Qwen2-VL-2B-Instruct dims (random-init bf16), batch=1 per GPU, one 448x448 image "
                                   "(1024 patches -> 256 image tokens) + 128 text tokens, greedy 256-token decode, EOS disabled",
                       "prompt_tokens": int(req[0].shape[1]), "max_tokens": args.max_tokens, "parallelism": f"dp{ws}",
                       "decode_lookahead": args.lookahead},
            "decode_us_per_token": us_per_token,
            "prefill_ms_to_first_token": pre_max / args.steps * 1e3,
            "prompt_tps": ws * args.steps * int(req[0].shape[1]) / pre_max,
            "e2e_tokens_per_s": ws * ntok / wall,
            "load_s": load_s,
            "roofline_decode_step": {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": step_gbs / HBM_PEAK_GBS, "traffic": traffic_tok,
                                     "algorithmic_bytes_per_token": bytes_per_token},
        }
        if extras:
            k = extras["kernels"]["gemv_gate_up_swiglu"]
            out["roofline"] = {"bound": "hbm", "kernel": GATE_UP_KERNEL + " (RMSNorm + gate/up GEMV + SwiGLU, 28 launches/token)",
                               "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["GBps"] / HBM_PEAK_GBS,
                               "traffic": traffic_gu, "bytes_per_launch": k["bytes_per_launch"], "us_per_launch": k["us_per_launch"]}
            out["kernel_rooflines"] = extras["kernels"]
            ips336, dt336 = extras["vit336"]
            ips448, dt448 = extras["vit448"]
            out["vision_images_per_s"] = ips336
            out["roofline_vit"] = {"bound": "mfma", "achieved": ips336 * VIT_TFLOP_336, "peak": MFMA_BF16_PEAK_TF,
                                   "unit": "TFLOP/s", "frac": ips336 * VIT_TFLOP_336 / MFMA_BF16_PEAK_TF, "traffic": None,
                                   "workload": f"{args.vit_batch} x 336x336 images per call ({args.vit_batch * 576} patches)",
                                   "ms_per_call": dt336 * 1e3}
            out["batch8_decode"] = extras["batch8"]
            out["continuous_batching"] = extras["continuous"]
            out["vision_single_448_images_per_s"] = ips448
            out["vision_single_448_tflops"] = ips448 * VIT_TFLOP_448
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    parallel.barrier()             # ranks leave together (rank 0 was still measuring the per-kernel rooflines)
    parallel.shutdown()


if __name__ == "__main__":
    main()
