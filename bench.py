#!/usr/bin/env python
"""Headline benchmark: decode tokens/sec + vision-prefill images/sec, Qwen2-VL-2B bf16 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload qwen2vl-2b | nanollava | qwen2vl-7b-b32 | qwen2vl-2b-w4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`): the default is configs[1] (below); `nanollava` is configs[0] (nanoLLaVA dims, one
336x336 image resized to 384x384, greedy 64 tokens); `qwen2vl-7b-b32` is configs[2] (Qwen2-VL-7B dims, 32 requests
dealt data-parallel over the ranks through parallel.dp_batch_generate, every rank a continuous BatchGenerator).

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
T_PROCESS_START = time.perf_counter()

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak
MFMA_RANDOM_OPERAND_TF = 1870.0   # measured (profiles/r06_gemm_power_limit.txt): bare v_mfma_f32_16x16x32_bf16 on N(0,1) operands, all CUs,
                                  # power-limited at 1.85 GHz (2.37 PF on zero operands) - an information line, never the roofline peak
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


def sampled_decode_throughput(model, req, max_tokens, lookahead):
    """The same request decoded with a SAMPLER in the captured step instead of the greedy tail (reference make_sampler,
    sample_utils.py:10-89: temperature 0.7 alone, + top_p 0.9, + the top-p / min-p / top-k chain) - what csrc/sample.hip's
    filter + Gumbel launches add to a decode step.  One warm pass, one timed pass each."""
    from mlx_vlm_amd.generate import generate_step

    ids, pix, thw = req
    out = {}
    for name, kw in (("temperature_0.7", {}), ("top_p_0.9", dict(top_p=0.9)), ("top_p_0.9_min_p_0.02_top_k_50", dict(top_p=0.9, min_p=0.02, top_k=50))):
        dec = 0.0
        for rep in range(2):
            gen = generate_step(ids, model, pix, None, max_tokens=max_tokens, temperature=0.7, seed=1234, image_grid_thw=thw,
                                return_logprobs=False, lookahead=lookahead, **kw)
            n, t_first = 0, None
            for _tok, _ in gen:
                if t_first is None:
                    t_first = time.perf_counter()
                n += 1
            torch.cuda.synchronize()
            dec = time.perf_counter() - t_first
        out[name] = {"generation_tps": (n - 1) / dec, "decode_us_per_token": dec / (n - 1) * 1e6}
    return out


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
        # timed as ONE captured graph of the launches (HIP events around 6 replays): a Python loop of ctypes calls is
        # host-dispatch bound below ~10 us per launch (round 3 reported 9.6 us for a 6.1 us kernel this way)
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        dt = time_events(g.replay, 6) / per
        res[name] = {"bytes_per_launch": nbytes, "us_per_launch": dt * 1e6, "GBps": nbytes / dt / 1e9,
                     "timing": f"hipGraph of {per} launch(es), HIP events over 6 replays"}
        del g
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


def wide_decode_throughput(model, cfg, rows=64, max_tokens=48):
    """Extra: `rows` concurrent requests (336x336 image + 128 text tokens each) through the continuous generator with that many
    decode rows - WIDE steps (17..64 rows: the prefill GEMMs + paged decode attention, engine.hip decode_impl); decode
    tokens/s of the generator's own clock (wall time with decode steps in flight)."""
    from mlx_vlm_amd import synthetic
    from mlx_vlm_amd.batch import generate_batch_continuous
    from mlx_vlm_amd.models import qwen2_vl

    # an engine of its own: 64 rows + the admissions prefilled ahead need 2 * rows + 2 sequence slots - a pool of that many
    # sequences would move the headline model from the identity to the paged KV layout
    del model
    dev = torch.device("cuda", torch.cuda.current_device())
    W = synthetic.random_weights(cfg, seed=0, device=dev)
    model = qwen2_vl.Model(cfg, device=dev, kv_pool_tokens=49152, max_seqs=2 * rows + 8)
    model.load_weights(W)
    del W
    reqs = [build_request(cfg, 336, 128, 900 + i) for i in range(rows)]
    ids = [r[0].reshape(-1) for r in reqs]
    pix = [r[1] for r in reqs]
    thw = [r[2] for r in reqs]
    generate_batch_continuous(model, ids, pix, thw, max_tokens=6, batch_size=rows)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks, st = generate_batch_continuous(model, ids, pix, thw, max_tokens=max_tokens, batch_size=rows)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"rows": rows, "image": "336x336", "max_tokens": max_tokens, "generation_tps": st.generation_tps,
            "decode_steps": st.decode_steps, "ms_per_step": 1e3 * st.generation_time / max(st.decode_steps, 1),
            "e2e_tokens_per_s": sum(len(t) for t in toks) / dt}


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


def cpu_baseline(threads, with_hf=True):
    """Reference-equivalent CPU path on the host cores, FULL model, nothing extrapolated: the oracle (torch-CPU
    restatement of the reference's bf16 typed graph; the reference itself needs `mlx`, which is not installable here)
    and, as the second opinion SURVEY section 8d asks for, HuggingFace `Qwen2VLForConditionalGeneration` in fp32 on
    torch-CPU.  Bounded sample (see `sample`): 6 decode tokens at context 384 through all 28 layers + lm_head, one
    448x448 and one 336x336 image through all 32 ViT blocks + merger."""
    from oracle import ops as O
    from oracle import qwen2_vl as oq

    torch.set_num_threads(threads)
    cfg = oq.Cfg()                                           # Qwen2-VL-2B dims
    t = cfg.text
    t0 = time.perf_counter()
    W = oq.random_weights(cfg, seed=0, dtype=torch.bfloat16)
    setup_s = time.perf_counter() - t0
    ctx, n_tok = 384, 6
    hd = t.hidden_size // t.num_attention_heads
    cache = [O.KVCache() for _ in range(t.num_hidden_layers)]
    g = torch.Generator().manual_seed(1)
    for c in cache:     # a context of 384 tokens already in the cache (contents random: timing only)
        c.update_and_fetch((torch.randn(1, t.num_key_value_heads, ctx, hd, generator=g) * 0.5).to(torch.bfloat16),
                           (torch.randn(1, t.num_key_value_heads, ctx, hd, generator=g) * 0.5).to(torch.bfloat16))
    e1 = (torch.randn(1, 1, t.hidden_size, generator=g) * 0.02).to(torch.bfloat16)
    oq.lm_head(W, cfg, oq.qwen2_model(W, cfg, e1, cache, torch.full((3, 1, 1), ctx)))     # warm-up token
    t0 = time.perf_counter()
    for i in range(n_tok):
        h = oq.qwen2_model(W, cfg, e1, cache, torch.full((3, 1, 1), ctx + 1 + i))
        O.argmax_first(O.logprobs_from_logits(oq.lm_head(W, cfg, h)[:, -1, :]))
    tok_s = n_tok / (time.perf_counter() - t0)
    img_s = {}
    for hw, n in ((448, 1024), (336, 576)):
        grid = np.array([[1, hw // 14, hw // 14]])
        pix = torch.randn(n, 1176, generator=g).to(torch.bfloat16)
        t0 = time.perf_counter()
        oq.vision_tower(W, cfg, pix, grid)
        img_s[hw] = 1.0 / (time.perf_counter() - t0)
    out = {"value": tok_s, "unit": "tokens/s", "cores": threads, "kind": "port",
           "vision_images_per_s": img_s[336], "vision_images_per_s_448": img_s[448],
           "sample": (f"oracle (torch-CPU restatement of the reference's bf16 graph), Qwen2-VL-2B at full size, {threads} threads: "
                      f"decode = {n_tok} tokens at context {ctx} through all {t.num_hidden_layers} layers + lm_head + greedy "
                      f"sampling; vision = one 336x336 image (576 patches) and one 448x448 image (1024 patches) through all "
                      f"{cfg.vision.depth} ViT blocks + merger; nothing extrapolated"),
           "setup_s": setup_s}
    del W, cache
    if with_hf:
        try:
            out["hf_fp32"] = _hf_cpu_baseline(cfg, threads, ctx, n_tok)
        except Exception as e:                       # the second opinion must never cost the headline line
            out["hf_fp32"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _hf_cpu_baseline(cfg, threads, ctx, n_tok):
    """HuggingFace transformers Qwen2VLForConditionalGeneration, fp32, torch-CPU, random init at the same dims: decode
    tokens/s at the same context (greedy, KV cache) and the vision tower on one 336x336 image."""
    import transformers
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration

    t, v = cfg.text, cfg.vision
    hcfg = Qwen2VLConfig(
        text_config=dict(hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
                         intermediate_size=t.intermediate_size, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps,
                         rope_theta=t.rope_theta, rope_scaling={"type": "mrope", "mrope_section": list(t.mrope_section)},
                         tie_word_embeddings=t.tie_word_embeddings, max_position_embeddings=32768, bos_token_id=0,
                         eos_token_id=1, pad_token_id=2),
        vision_config=dict(depth=v.depth, embed_dim=v.embed_dim, hidden_size=v.hidden_size, num_heads=v.num_heads,
                           mlp_ratio=int(v.mlp_ratio), patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
                           temporal_patch_size=v.temporal_patch_size, in_channels=v.in_channels),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
        vision_start_token_id=cfg.vision_start_token_id, tie_word_embeddings=t.tie_word_embeddings, bos_token_id=0,
        eos_token_id=1, pad_token_id=2)
    t0 = time.perf_counter()
    with torch.no_grad():
        m = Qwen2VLForConditionalGeneration(hcfg).eval().to(torch.float32)     # HF's own random initialisation
        setup = time.perf_counter() - t0
        ids = torch.randint(3, min(t.vocab_size, 151643) - 8, (1, ctx))
        out = m(input_ids=ids, use_cache=True)
        past = out.past_key_values
        nxt = out.logits[:, -1].argmax(-1, keepdim=True)
        out = m(input_ids=nxt, past_key_values=past, use_cache=True)          # warm-up token
        past, nxt = out.past_key_values, out.logits[:, -1].argmax(-1, keepdim=True)
        t0 = time.perf_counter()
        for _ in range(n_tok):
            out = m(input_ids=nxt, past_key_values=past, use_cache=True)
            past, nxt = out.past_key_values, out.logits[:, -1].argmax(-1, keepdim=True)
        tok_s = n_tok / (time.perf_counter() - t0)
        pix = torch.randn(576, 1176)
        grid = torch.tensor([[1, 24, 24]])
        visual = m.model.visual if hasattr(m, "model") and hasattr(m.model, "visual") else m.visual
        t0 = time.perf_counter()
        visual(pix, grid_thw=grid)
        img_s = 1.0 / (time.perf_counter() - t0)
    return {"value": tok_s, "unit": "tokens/s", "vision_images_per_s": img_s, "cores": threads, "setup_s": setup,
            "sample": f"transformers {transformers.__version__} Qwen2VLForConditionalGeneration fp32, random init, "
                      f"{n_tok} greedy tokens at context {ctx} with its KV cache; vision tower on one 336x336 image"}


DECODE_KERNELS = ("gemv_rowwave_kernel", "gemv_splitk_kernel", "attn_decode_mfma_kernel", "attn_decode_pagesplit_kernel",
                  "attn_decode_combine_kernel", "lse_partial_kernel", "logprob_argmax_kernel", "argmax_final_kernel",
                  "embed_gather_kernel", "decode_advance_kernel", "sample_filter_kernel", "logprob_argmax_tail_kernel")
HEAD_KERNEL = "gemv_rowwave_kernel<4, 3, 1, 1, 0>"         # RMSNorm + lm_head GEMV: exactly one launch per decoded token
GATE_UP_KERNEL = "gemv_rowwave_kernel<4, 3, 1, 1, 16>"     # name as rocprofv3 prints it (R=4 rows/wave, RMSNorm prologue, SwiGLU)
DECODE_CSRC = ("gemv_bf16.hip", "attn_decode.hip", "attn_pagesplit.cuh", "sample.hip", "embed.hip", "engine.hip", "common.cuh",
               "internal.h")


def decode_csrc_sha16():
    """hash of the sources of every kernel in the decode step + the engine that sequences them: what a PMC pass was taken on"""
    import hashlib

    h = hashlib.sha256()
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mlx-vlm_amd", "csrc")
    for f in DECODE_CSRC:
        h.update(open(os.path.join(base, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per launch from the committed --pmc passes (scripts/r04_final.sh -> profiles/r04_pmc_traffic.json):
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of MI355X_MICROARCH.md.  bench.py cannot collect hardware
    counters itself (they need rocprofv3 around the process).  The file records the hash of the decode step's kernel sources
    it was taken on (`_meta.decode_csrc_sha16`, scripts/pmc_summary.py); a file without it or with another hash is STALE and
    refused: -> (None, None, reason)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    cands = sorted((f for f in os.listdir(here) if f.endswith("_pmc_traffic.json")), reverse=True) if os.path.isdir(here) else []
    if not cands:
        return None, None, "no profiles/*_pmc_traffic.json"
    path = os.path.join(here, cands[0])
    d = json.load(open(path))
    sha = d.get("_meta", {}).get("decode_csrc_sha16")
    if sha != decode_csrc_sha16():
        return None, None, (f"{cands[0]} is stale: taken on decode sources {sha}, this tree is {decode_csrc_sha16()} "
                            "(re-run scripts/r04_final.sh)")
    gu = d.get(GATE_UP_KERNEL, {}).get("hbm_bytes_per_launch")
    steps = d.get(HEAD_KERNEL, {}).get("launches", 0) or d.get("decode_advance_kernel", {}).get("launches", 0)
    per_tok = None
    if steps:
        per_tok = sum(v.get("hbm_bytes_per_launch", 0.0) * v["launches"] for k, v in d.items()
                      if k.startswith(DECODE_KERNELS)) / steps
    return gu, per_tok, cands[0]


def _load_synthetic(cfg_dict, model_pkg, rank, dev, w4=False, **engine_kw):
    """rank 0 materialises the synthetic replica, the others receive it over RCCL/xGMI (parallel.broadcast_weights).
    w4: the language model as an MLX affine 4-bit checkpoint (random nibbles / scales / biases of that layout)."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.utils import fit_host_threads, freeze_heap

    cfg = model_pkg.ModelConfig.from_dict(dict(cfg_dict))
    t0 = time.perf_counter()
    W = synthetic.random_weights(cfg, seed=0, device=dev, fill=(rank == 0))
    if w4 and getattr(cfg, "model_type", "") == "phi3_v":
        synthetic.quantize_random_(W, prefix="", skip=("model.vision_embed_tokens.",))
    elif w4:
        synthetic.quantize_random_(W)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    parallel.broadcast_weights(W, src=0)
    torch.cuda.synchronize()
    bcast_s = parallel.max_over_ranks(time.perf_counter() - t1, dev)
    nbytes = sum(v.numel() * v.element_size() for v in W.values())
    model = model_pkg.Model(cfg, device=dev, **engine_kw)
    model.load_weights(W)
    del W
    torch.cuda.synchronize()
    freeze_heap()          # what load() does: no 100 ms cyclic-GC passes over the import heap inside timed loops
    fit_host_threads()     # what load() does: torch's CPU pool capped at the container's CPU quota
    return cfg, model, {"load_s": time.perf_counter() - t0, "weight_bytes": nbytes, "broadcast_s": bcast_s}


def _dist_info(ws, load=None):
    import torch.distributed as dist

    out = {"backend": None, "ranks": 1}
    if ws > 1 and dist.is_initialized():
        out = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}
        if load and load.get("broadcast_s"):
            out["weight_broadcast_GBps"] = load["weight_bytes"] / load["broadcast_s"] / 1e9
    return out


def workload_nanollava(args, rank, ws, dev):
    """BASELINE configs[0]: nanoLLaVA (SigLIP-so400m/14-384 + Qwen1.5-0.5B), one 336x336 image (resized to 384x384 -> 729
    image tokens) + 128 text tokens, greedy 64 tokens, batch 1 per GPU (weak scaling)."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.generate import generate_step
    from mlx_vlm_amd.models import llava_bunny

    cfg, model, load = _load_synthetic(synthetic.NANOLLAVA, llava_bunny, rank, dev, kv_pool_tokens=8192, max_seqs=8)
    max_tokens = args.max_tokens or 64
    rng = np.random.default_rng(rank)
    img = rng.integers(0, 256, (336, 336, 3), dtype=np.uint8)
    pix = torch.from_numpy(np.stack(llava_bunny.ImageProcessor().preprocess([img]))).to(dev)
    text = np.random.default_rng(1000 + rank).integers(0, 151643, 128)
    ids = np.concatenate([text[:64], [cfg.image_token_index], text[64:]]).astype(np.int64)[None]

    def step():
        t0 = time.perf_counter()
        toks, t_first = [], None
        for tok, _ in generate_step(ids, model, pix, None, max_tokens=max_tokens, temperature=0.0, return_logprobs=False,
                                    lookahead=args.lookahead):
            if t_first is None:
                t_first = time.perf_counter()
            toks.append(tok)
        torch.cuda.synchronize()
        return t_first - t0, time.perf_counter() - t_first, toks

    for _ in range(args.warmup):
        step()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pre = dec = 0.0
    for _ in range(args.steps):
        a, b, _ = step()
        pre, dec = pre + a, dec + b
    torch.cuda.synchronize()
    parallel.barrier()
    wall = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    dec_max, pre_max = parallel.max_over_ranks(dec, dev), parallel.max_over_ranks(pre, dev)
    n_dec = args.steps * (max_tokens - 1)
    tps = ws * n_dec / dec_max
    t, v = cfg.text_config, cfg.vision_config
    per_layer = 4 * t.hidden_size * t.hidden_size + 3 * t.hidden_size * t.intermediate_size
    prompt_tokens = ids.shape[1] - 1 + model.vision_tower.num_patches
    kv_per_tok = 2 * t.num_hidden_layers * t.num_key_value_heads * (t.hidden_size // t.num_attention_heads) * 2
    bytes_per_token = 2 * (t.num_hidden_layers * per_layer + t.vocab_size * t.hidden_size) + kv_per_tok * (prompt_tokens + max_tokens // 2)
    us_tok = dec_max / n_dec * 1e6
    out = {"metric": "decode tokens/sec + vision-prefill images/sec, nanoLLaVA", "value": tps, "unit": "tokens/s", "n_gpus": ws,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "nanoLLaVA dims (Qwen1.5-0.5B + SigLIP-so400m/14-384, random-init bf16), batch=1 per GPU, one "
                                  "336x336 image resized to 384x384 (729 image tokens) + 128 text tokens, greedy decode, EOS disabled",
                      "prompt_tokens": int(prompt_tokens), "max_tokens": max_tokens, "parallelism": f"dp{ws}"},
           "decode_us_per_token": us_tok, "prefill_ms_to_first_token": pre_max / args.steps * 1e3,
           "prompt_tps": ws * prompt_tokens * args.steps / pre_max, "load": load, "distributed": _dist_info(ws, load),
           "roofline": {"bound": "hbm", "kernel": "whole decode step", "achieved": bytes_per_token / us_tok * 1e-3,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_per_token / us_tok * 1e-3 / HBM_PEAK_GBS,
                        "traffic": None, "algorithmic_bytes_per_token": bytes_per_token}}
    if rank == 0 and not args.no_extras:
        N, E, I = model.vision_tower.num_patches, v.hidden_size, v.intermediate_size
        tflop = (v.num_hidden_layers * (2 * N * E * 3 * E + 2 * N * E * E + 4 * N * E * I + 4 * N * N * E)
                 + 2 * N * model.vision_tower.patch_dim * E + 2 * N * (E * t.hidden_size + t.hidden_size ** 2)) / 1e12

        def tower(n):
            batch = pix.expand(n, -1, -1, -1).contiguous()
            for _ in range(2):
                model.encode_image(batch)
            torch.cuda.synchronize()
            dts = sorted(time_events(lambda: model.encode_image(batch), 1) for _ in range(5))
            return n / dts[2]

        ips1, ips8 = tower(1), tower(8)
        out["vision_images_per_s"] = ips8
        out["vision_images_per_s_single"] = ips1
        out["roofline_vit"] = {"bound": "mfma", "achieved": ips8 * tflop, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                               "frac": ips8 * tflop / MFMA_BF16_PEAK_TF, "tflop_per_image": tflop, "traffic": None,
                               "workload": "8 x 384x384 images per call (SigLIP tower + projector)"}
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        from mlx_vlm_amd.utils import cpu_quota
        out["cpu_baseline"] = cpu_baseline_nanollava(min(cpu_quota(), 32))
    return out


def cpu_baseline_nanollava(threads):
    """The oracle (torch-CPU restatement of the reference's llava_bunny files, bf16 typed graph) at full size on the host
    cores: 6 decode tokens at context 857 through all 24 layers + lm_head, one image through the 27-layer tower."""
    from oracle import llava_bunny as ob
    from oracle import ops as O

    torch.set_num_threads(threads)
    cfg = ob.Cfg(text=ob.TextCfg(), vision=ob.VisionCfg())
    W = ob.random_weights(cfg, seed=0, dtype=torch.bfloat16, std=0.02, embed_std=0.02)
    t = cfg.text
    ctx, n_tok, hd = 857, 6, t.hidden_size // t.num_attention_heads
    g = torch.Generator().manual_seed(1)
    cache = [O.KVCache() for _ in range(t.num_hidden_layers)]
    for c in cache:
        c.update_and_fetch((torch.randn(1, t.num_key_value_heads, ctx, hd, generator=g) * 0.5).to(torch.bfloat16),
                           (torch.randn(1, t.num_key_value_heads, ctx, hd, generator=g) * 0.5).to(torch.bfloat16))
    ob.decode_teacher_forced  # noqa: B018  (same code path, kept importable)
    e1 = (torch.randn(1, 1, t.hidden_size, generator=g) * 0.02).to(torch.bfloat16)

    def one(e):
        h = e
        for i in range(t.num_hidden_layers):
            h = ob.decoder_layer(W, i, cfg, h, cache[i])
        h = O.rms_norm(h, W[ob.LM + "norm.weight"], t.rms_norm_eps)
        return O.argmax_first(O.logprobs_from_logits(O.linear(h, W[ob.LM + "embed_tokens.weight"])[:, -1, :]))

    one(e1)
    t0 = time.perf_counter()
    for _ in range(n_tok):
        one(e1)
    tok_s = n_tok / (time.perf_counter() - t0)
    pix = torch.randn(1, 3, 384, 384, generator=g).to(torch.bfloat16)
    t0 = time.perf_counter()
    ob.mm_projector(W, ob.vision_tower(W, cfg, pix))
    img_s = 1.0 / (time.perf_counter() - t0)
    return {"value": tok_s, "unit": "tokens/s", "cores": threads, "kind": "port", "vision_images_per_s": img_s,
            "sample": f"oracle (torch-CPU restatement of the reference, bf16 graph), nanoLLaVA at full size, {threads} threads: "
                      f"{n_tok} decode tokens at context {ctx} through all {t.num_hidden_layers} layers + lm_head; one 384x384 "
                      f"image through the 27-layer SigLIP tower + projector; nothing extrapolated"}


def cpu_baseline_lm(kind, threads, short=False):
    """cpu_baseline of the non-headline workloads: the oracle (torch-CPU restatement of the reference's graph for that model
    family) at FULL size on the host cores, single-stream decode - a bounded sample (3 or 6 tokens at the workload's context,
    all layers + lm_head + greedy sampling; K / V of the context pre-filled with random values: timing only).  The big
    matrices of the synthetic checkpoint come from oracle.ops.fast_normal (seconds instead of minutes of setup).
    short (the `configs` block of the default line): 2 timed tokens, and 4-bit checkpoints take RANDOM packed words instead
    of quantising billions of weights on the host (timing only) - the sample says so."""
    from oracle import ops as O

    torch.set_num_threads(threads)
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    t0 = time.perf_counter()
    if kind in ("qwen2vl-7b", "qwen2vl-2b-w4"):
        from oracle import qwen2_vl as oq
        if kind == "qwen2vl-7b":
            cfg = oq.Cfg(text=oq.TextCfg(hidden_size=3584, num_hidden_layers=28, intermediate_size=18944, num_attention_heads=28,
                                         num_key_value_heads=4, vocab_size=152064, tie_word_embeddings=False),
                         vision=oq.VisionCfg(depth=1, embed_dim=1280, hidden_size=3584, num_heads=16))
            ctx, n_tok, label = 274, 3, "Qwen2-VL-7B language model (28 layers of 3584 / 18944, untied head)"
        else:
            cfg = oq.Cfg(vision=oq.VisionCfg(depth=1))
            ctx, n_tok, label = 386, 6, "Qwen2-VL-2B language model as an MLX affine 4-bit checkpoint (oracle/quant.py)"
        W = oq.random_weights(cfg, seed=0, dtype=BF, fast=True)
        if kind == "qwen2vl-2b-w4":
            from oracle import quant as Q
            W = (_fast_w4_oracle_weights(W, lambda p, v: p.startswith("language_model.")) if short else
                 Q.quantize_checkpoint(W, predicate=lambda p, v: p.startswith("language_model."))[1])
        t = cfg.text
        hd, nkv, nl = t.hidden_size // t.num_attention_heads, t.num_key_value_heads, t.num_hidden_layers
        step = lambda e, cache, i: O.argmax_first(O.logprobs_from_logits(  # noqa: E731
            oq.lm_head(W, cfg, oq.qwen2_model(W, cfg, e, cache, torch.full((3, 1, 1), ctx + i)))[:, -1, :]))
    elif kind == "idefics2-8b":
        from oracle import idefics2 as om
        cfg = om.Cfg(text=om.TextCfg(), vision=om.VisionCfg(num_hidden_layers=1), perceiver=om.PerceiverCfg())
        W = om.random_weights(cfg, seed=0, dtype=BF, std=0.02, embed_std=0.02, fast=True)
        t = cfg.text
        hd, nkv, nl = 128, t.num_key_value_heads, t.num_hidden_layers
        ctx, n_tok, label = 384, 3, "Idefics2-8B language model (Mistral-7B: 32 layers of 4096 / 14336)"
        step = lambda e, cache, i: O.argmax_first(O.logprobs_from_logits(om.language_model(W, cfg, e, cache, last_only=True)[:, -1, :]))  # noqa: E731
    elif kind == "phi35v-w4":
        from oracle import phi3_v as om
        from oracle import quant as Q
        short, long = om.su_factors(96, seed=9)
        cfg = om.Cfg(text=om.TextCfg(short_factor=short, long_factor=long), vision=om.VisionCfg(num_hidden_layers=1))
        W = om.random_weights(cfg, seed=0, dtype=BF, std=0.02, embed_std=0.02, fast=True)
        W = (_fast_w4_oracle_weights(W, lambda p, v: not p.startswith("model.vision_embed_tokens.")) if short else
             Q.quantize_checkpoint(W, predicate=lambda p, v: not p.startswith("model.vision_embed_tokens."))[1])
        t = cfg.text
        hd, nkv, nl = t.hidden_size // t.num_attention_heads, t.num_key_value_heads, t.num_hidden_layers
        ctx, n_tok, label = 885, 6, "Phi-3.5-vision language model (32 layers of 3072 / 8192) as an MLX affine 4-bit checkpoint"
        step = lambda e, cache, i: O.argmax_first(O.logprobs_from_logits(om.language_model(W, cfg, e, cache, last_only=True)[:, -1, :]))  # noqa: E731
    else:
        raise ValueError(kind)
    setup_s = time.perf_counter() - t0
    cache = [O.KVCache() for _ in range(nl)]
    for c in cache:
        c.update_and_fetch((torch.randn(1, nkv, ctx, hd, generator=g) * 0.5).to(BF), (torch.randn(1, nkv, ctx, hd, generator=g) * 0.5).to(BF))
    e1 = (torch.randn(1, 1, t.hidden_size, generator=g) * 0.02).to(BF)
    if short:
        n_tok = 2
    step(e1, cache, 0)                                        # warm-up token
    t0 = time.perf_counter()
    for i in range(n_tok):
        step(e1, cache, 1 + i)
    tok_s = n_tok / (time.perf_counter() - t0)
    return {"value": tok_s, "unit": "tokens/s", "cores": threads, "kind": "port", "setup_s": setup_s,
            "sample": f"oracle (torch-CPU restatement of the reference's typed graph), {label} at full size, {threads} threads: "
                      f"{n_tok} single-stream decode tokens at context {ctx} through all {nl} layers + lm_head + greedy sampling; "
                      "nothing extrapolated (the CPU path has no batched step: one sequence)"
                      + ("; 4-bit matrices hold random packed words (timing only)" if short and "w4" in kind else "")}


def _with_cpu_baseline(out, kind, args, rank, ws):
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        from mlx_vlm_amd.utils import cpu_quota
        try:
            out["cpu_baseline"] = cpu_baseline_lm(kind, min(cpu_quota(), 32))
        except Exception as e:                                # the baseline leg must never cost the measured line
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def workload_2b_w4(args, rank, ws, dev):
    """the headline workload over an MLX affine 4-bit language model (what the reference's README runs:
    Qwen2-VL-2B-Instruct-4bit) - bf16 activations / KV / vision tower, 4-bit + group-64 scale / bias weights in the decoder,
    embedding and head.  Decode through the dequant-fused GEMVs (csrc/gemv_w4.hip)."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.models import qwen2_vl

    cfg, model, load = _load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, rank, dev, w4=True, kv_pool_tokens=32768, max_seqs=40)
    max_tokens = args.max_tokens or 256
    req = build_request(cfg, 448, 128, seed=rank)
    req = (req[0], req[1].to(dev), req[2])
    for _ in range(args.warmup):
        run_step(model, req, max_tokens, args.lookahead)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pre = dec = 0.0
    for _ in range(args.steps):
        a, b, _ = run_step(model, req, max_tokens, args.lookahead)
        pre, dec = pre + a, dec + b
    torch.cuda.synchronize()
    parallel.barrier()
    wall = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    dec_max, pre_max = parallel.max_over_ranks(dec, dev), parallel.max_over_ranks(pre, dev)
    n_dec = args.steps * (max_tokens - 1)
    us_tok = dec_max / n_dec * 1e6
    lm_params = 28 * 46797824 + 233373696                       # decoder Linears + the tied head, read once per token
    ctx_mid = int(req[0].shape[1]) + max_tokens // 2
    bytes_per_token = lm_params * 9 // 16 + 1536 * 2 + 28672 * ctx_mid + 28672     # 4 bits + 32 / 64 bits per weight
    extras = {}
    if ws == 1 and not args.no_extras:      # batched steps: 8 rows on the v_dot2c 4-bit GEMVs, 16 on the dequant-fused MFMA form
        extras = {"batch8_decode": batch_decode_throughput(model, cfg, 8), "batch16_decode": batch_decode_throughput(model, cfg, 16)}
    return {"extras": extras, "metric": "decode tokens/sec, Qwen2-VL-2B 4-bit (MLX affine, group 64)", "value": ws * n_dec / dec_max, "unit": "tokens/s",
            "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 activations, int4 affine weights (fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "Qwen2-VL-2B-Instruct dims, language model as an MLX affine 4-bit checkpoint (random nibbles / "
                                   "scales / biases), batch=1 per GPU, one 448x448 image + 128 text tokens, greedy "
                                   f"{max_tokens}-token decode, EOS disabled",
                       "prompt_tokens": int(req[0].shape[1]), "max_tokens": max_tokens, "parallelism": f"dp{ws}"},
            "ttft_ms": pre_max / args.steps * 1e3, "us_per_token": us_tok, "load": load, "distributed": _dist_info(ws, load),
            "roofline": {"bound": "hbm", "kernel": "whole decode step (4-bit weights + bf16 KV)", "achieved": bytes_per_token / us_tok / 1e3,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_per_token / us_tok / 1e3 / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_token": bytes_per_token}}


def workload_7b_b32(args, rank, ws, dev):
    """BASELINE configs[2]: Qwen2-VL-7B dims, 32 requests (336x336 image + 128-token prompt each) dealt data-parallel over
    the ranks by parallel.dp_batch_generate (length-sorted deal; every rank a continuous BatchGenerator of up to 32 rows -
    VLM_BENCH_7B_ROWS; no collective inside the steps).  On one GPU the 32 requests decode in one wave of 32-row WIDE
    steps (prefill GEMMs + paged decode attention); a rank with <= 16 requests runs the 16-row MFMA decode GEMM steps.
    Total work is fixed: strong scaling."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.models import qwen2_vl

    # up to 32 decode rows per GPU: with one rank the 32 requests decode in ONE wave of wide steps (prefill GEMMs, engine.hip
    # decode_impl); dealt over more ranks a rank's 16 / 8 / 4 requests run the 16-row (or narrower) GEMV steps
    rows = int(os.environ.get("VLM_BENCH_7B_ROWS", "32"))
    cfg, model, load = _load_synthetic(synthetic.QWEN2_VL_7B, qwen2_vl, rank, dev, kv_pool_tokens=32768, max_seqs=2 * rows + 8)
    n_req, max_tokens = 32, args.max_tokens or 64
    reqs = []
    for i in range(n_req):
        ids, pix, thw = build_request(cfg, 336, 128, seed=i)
        reqs.append({"input_ids": ids.reshape(-1), "pixel_values": pix, "image_grid_thw": thw, "max_tokens": max_tokens})
    for _ in range(args.warmup):
        parallel.dp_batch_generate(model, None, requests=reqs[: 2 * ws], max_tokens=8, batch_size=rows)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total, res = 0, None
    dec_tok = dec_steps = 0
    dec_t = 0.0
    for _ in range(args.steps):
        res = parallel.dp_batch_generate(model, None, requests=reqs, max_tokens=max_tokens, batch_size=rows)
        if rank == 0:
            total += res["generation_tokens"]
            dec_tok, dec_steps, dec_t = dec_tok + res["decode_tokens"], dec_steps + res["decode_steps"], dec_t + res["decode_time_s"]
    torch.cuda.synchronize()
    parallel.barrier()
    wall = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    t = cfg.text_config
    lm_params = t.num_hidden_layers * (2 * t.hidden_size * (t.num_attention_heads + t.num_key_value_heads) * 128
                                       + 3 * t.hidden_size * t.intermediate_size) + t.vocab_size * t.hidden_size
    out = {"metric": "decode tokens/sec (end to end, prefill included), Qwen2-VL-7B batch=32", "value": total / wall if rank == 0 else 0.0,
           "unit": "tokens/s", "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "Qwen2-VL-7B-Instruct dims (random-init bf16), 32 requests (one 336x336 image -> 144 image tokens + "
                                  f"128 text tokens each, greedy {max_tokens} new tokens, EOS disabled) dealt data-parallel over the "
                                  f"ranks, continuous batching with up to {rows} decode rows per GPU (8 / 16-row steps: skinny-M MFMA decode GEMM; "
                                  "32-row steps: the prefill GEMMs + paged decode attention)",
                      "requests": n_req, "max_tokens": max_tokens, "parallelism": f"dp{ws}",
                      "per_rank_requests": res["per_rank_requests"] if rank == 0 else None},
           "load": load, "distributed": _dist_info(ws, load), "decode_rows_per_gpu": rows,
           }
    if rank == 0:
        # decode steps of the job (graph replays summed over the ranks; decode time = the slowest rank's wall time with decode
        # steps in flight, BatchGenerator.stats().generation_time - the prefills admitted UNDER those steps are inside it):
        # a step streams the weights once and, per row it serves, that row's K / V (57,344 B per cached token at 7B) at
        # the mean context
        kv_tok = 2 * t.num_hidden_layers * t.num_key_value_heads * 128 * 2
        ctx_mid = int(reqs[0]["input_ids"].size) + max_tokens // 2
        job_bytes = dec_steps * 2 * lm_params + dec_tok * ctx_mid * kv_tok
        gbs = job_bytes / max(dec_t, 1e-9) / 1e9 / ws            # per GPU: the ranks' steps run concurrently
        out["decode_tokens_per_s"] = dec_tok / max(dec_t, 1e-9)
        out["roofline"] = {"bound": "hbm", "kernel": "decode steps of the job (bf16 weights once per step + K / V of the rows it serves), per GPU",
                           "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_weight_bytes_per_step": 2 * lm_params, "decode_steps": dec_steps,
                           "decode_tokens": dec_tok, "decode_time_s": dec_t, "kv_bytes_per_cached_token": kv_tok}
    return out


def workload_idefics2_b8(args, rank, ws, dev):
    """BASELINE configs[3]: Idefics2-8B (SigLIP-so400m tower + perceiver resampler + Mistral-7B, bf16), multi-image prompts -
    4 x 336 x 336 images per prompt (378 x 378 after the processor's resize rule: 729 patches each -> 4 x 64 image tokens
    interleaved with the text) + 128 text tokens, batch 8 PER GPU through the continuous generator, greedy 64 new tokens."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.batch import generate_batch_continuous
    from mlx_vlm_amd.models import idefics2

    cfg, model, load = _load_synthetic(synthetic.IDEFICS2_8B, idefics2, rank, dev, kv_pool_tokens=32768, max_seqs=40)
    n_req, max_tokens, n_img = 8, args.max_tokens or 64, 4
    ip = idefics2.Idefics2ImageProcessor()
    nl = cfg.perceiver_config.resampler_n_latents
    ids_l, pix_l, ex_l = [], [], []
    for i in range(n_req):
        rng = np.random.default_rng(1000 * rank + i)
        out = ip([[rng.integers(1, 256, (336, 336, 3), dtype=np.uint8) for _ in range(n_img)]])
        text = rng.integers(3, 32000, 128)
        parts = []
        for j in range(n_img):
            parts += [text[32 * j: 32 * (j + 1)], np.full(nl, cfg.image_token_id)]
        ids_l.append(np.concatenate(parts).astype(np.int64))
        pix_l.append(torch.from_numpy(out["pixel_values"]).to(dev))
        ex_l.append({"pixel_attention_mask": out["pixel_attention_mask"]})
    run = lambda n, mt: generate_batch_continuous(model, ids_l[:n], pix_l[:n], [None] * n, max_tokens=mt, extras=ex_l[:n],  # noqa: E731
                                                  batch_size=8)
    for _ in range(args.warmup):
        run(n_req, 8)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gen_tok = gen_t = pre_tok = pre_t = 0.0
    for _ in range(args.steps):
        toks, st = run(n_req, max_tokens)
        gen_tok, gen_t = gen_tok + st.generation_tokens, gen_t + st.generation_time
        pre_tok, pre_t = pre_tok + st.prompt_tokens, pre_t + st.prompt_time
    torch.cuda.synchronize()
    parallel.barrier()
    wall = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    gen_t_max = parallel.max_over_ranks(gen_t, dev)
    t = cfg.text_config
    D, TI = t.hidden_size, t.intermediate_size
    lm_params = t.num_hidden_layers * (D * (t.num_attention_heads + 2 * t.num_key_value_heads) * 128 + D * D + 3 * D * TI) + t.vocab_size * D
    kv_tok = 2 * t.num_hidden_layers * t.num_key_value_heads * 128 * 2
    ctx_mid = int(ids_l[0].size) + max_tokens // 2
    step_bytes = 2 * lm_params + n_req * ctx_mid * kv_tok
    steps_per_s = gen_tok / n_req / gen_t_max
    return {"metric": "decode tokens/sec, Idefics2-8B multi-image (4 x 336x336 per prompt), batch=8 per GPU", "value": ws * gen_tok / gen_t_max,
            "unit": "tokens/s", "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Idefics2-8B dims (SigLIP-so400m/14 tower 27 layers + perceiver resampler + Mistral-7B, random-init bf16), "
                                   "8 requests per GPU: 4 x 336x336 images (378x378 after resize, 729 patches -> 64 latents each) interleaved "
                                   f"with 128 text tokens, greedy {max_tokens} new tokens, EOS disabled, 8 decode rows",
                       "requests_per_gpu": n_req, "images_per_prompt": n_img, "prompt_tokens": int(ids_l[0].size),
                       "max_tokens": max_tokens, "parallelism": f"dp{ws}"},
            "e2e_tokens_per_s": ws * gen_tok / wall, "prompt_tps": ws * pre_tok / max(pre_t, 1e-9),
            "images_per_s_prefill": ws * n_req * n_img * args.steps / max(pre_t, 1e-9), "load": load, "distributed": _dist_info(ws, load),
            "roofline": {"bound": "hbm", "kernel": "whole 8-row decode step (bf16 weights once + 8 rows of K / V)",
                         "achieved": step_bytes * steps_per_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": step_bytes * steps_per_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_step": step_bytes}}


def workload_phi35v_w4_b16(args, rank, ws, dev):
    """BASELINE configs[4]: Phi-3.5-vision-instruct with an MLX affine 4-bit language model (the dequant-fused kernels:
    csrc/gemv_w4.hip at 1-4 rows, the W4 form of csrc/gemv_mfma.hip at 5-16 rows; prefill = the dequant-fused GEMM vlm_gemm_w4 up to 2048 rows per call, dequantise + the 256x256 bf16 GEMM beyond), batch
    16 PER GPU: 16 requests of one 336 x 336 image (HD transform at num_crops 4: 5 CLIP views -> 757 image tokens) + 128
    text tokens, greedy 64 new tokens, through the continuous generator at 16 decode rows (weak scaling: every rank
    serves its own 16)."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.batch import generate_batch_continuous
    from mlx_vlm_amd.models import phi3_v

    cfg, model, load = _load_synthetic(synthetic.PHI35_VISION, phi3_v, rank, dev, w4=True, kv_pool_tokens=32768, max_seqs=40)
    n_req, max_tokens = 16, args.max_tokens or 64
    ip = phi3_v.Phi3VImageProcessor()
    ids_l, pix_l, ex_l = [], [], []
    for i in range(n_req):
        rng = np.random.default_rng(1000 * rank + i)
        out = ip([rng.integers(0, 256, (336, 336, 3), dtype=np.uint8)])
        n_img = ip.calc_num_image_tokens(np.zeros((336, 336, 3), np.uint8))
        text = rng.integers(3, 32000, 128)
        ids_l.append(np.concatenate([text[:64], np.full(n_img, -1), text[64:]]).astype(np.int64))
        pix_l.append(torch.from_numpy(out["pixel_values"]).to(dev))
        ex_l.append({"image_sizes": out["image_sizes"]})
    kv_bits = args.kv_bits or None             # --kv-bits 8: the uniform 8-bit KV cache (QuantizedKVCache) for every row
    run = lambda n, mt: generate_batch_continuous(model, ids_l[:n], pix_l[:n], [None] * n, max_tokens=mt, extras=ex_l[:n],  # noqa: E731
                                                  kv_bits=kv_bits)
    for _ in range(args.warmup):
        run(n_req, 8)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gen_tok = gen_t = pre_tok = pre_t = 0.0
    for _ in range(args.steps):
        toks, st = run(n_req, max_tokens)
        gen_tok, gen_t = gen_tok + st.generation_tokens, gen_t + st.generation_time
        pre_tok, pre_t = pre_tok + st.prompt_tokens, pre_t + st.prompt_time
    torch.cuda.synchronize()
    parallel.barrier()
    wall = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    gen_t_max = parallel.max_over_ranks(gen_t, dev)
    D, I, H = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
    lm_params = cfg.num_hidden_layers * (3 * H * 128 * D + D * H * 128 + 3 * D * I) + 2 * cfg.vocab_size * D   # engine layout (96 -> 128)
    # per 16-row step: the weights once (4 bits + 32 / 64 bits per weight) + every row's K / V (MHA: 32 kv heads of 96 in
    # 32 layers = 393,216 B per cached token; the engine's 128-wide pages move 4 / 3 of that) at the mean context
    kv_tok = 2 * cfg.num_hidden_layers * cfg.num_key_value_heads * (D // H) * 2
    if kv_bits:                                # 8 bits + one (scale, bias) bf16 pair per 64 elements: 8.5 bits per element
        kv_tok = kv_tok * 17 // 32
    ctx_mid = int(ids_l[0].size) + max_tokens // 2
    step_bytes = lm_params * 9 // 16 + n_req * ctx_mid * kv_tok
    steps_per_s = gen_tok / n_req / gen_t_max
    v = cfg.vision_config
    N = 577
    clip_tflop = 5 * ((v.num_hidden_layers - 1) * (8 * N * v.hidden_size ** 2 + 4 * N * v.hidden_size * v.intermediate_size
                                                   + 4 * N * N * v.hidden_size) + 2 * 576 * 588 * v.hidden_size) / 1e12
    out = {"metric": "decode tokens/sec, Phi-3.5-vision int4 (MLX affine, group 64), batch=16 per GPU" + (", 8-bit KV cache" if kv_bits else ""),
           "kv_bits": kv_bits, "value": ws * gen_tok / gen_t_max,
           "unit": "tokens/s", "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16 activations, int4 affine weights (fp32 accumulate)", "data": "synthetic",
           "config": {"workload": "Phi-3.5-vision-instruct dims (Phi-3-mini 3.8B decoder as an MLX affine 4-bit checkpoint: random "
                                  "nibbles / scales / biases; CLIP ViT-L/14-336 bf16), 16 requests per GPU: one 336x336 image (5 views -> "
                                  f"757 image tokens) + 128 text tokens, greedy {max_tokens} new tokens, EOS disabled, 16 decode rows",
                      "requests_per_gpu": n_req, "prompt_tokens": int(ids_l[0].size), "max_tokens": max_tokens, "parallelism": f"dp{ws}"},
           "e2e_tokens_per_s": ws * gen_tok / wall, "prompt_tps": ws * pre_tok / max(pre_t, 1e-9),
           "images_per_s_prefill": ws * n_req * args.steps / max(pre_t, 1e-9), "clip_tflop_per_image": clip_tflop,
           "load": load, "distributed": _dist_info(ws, load),
           "roofline": {"bound": "hbm", "kernel": "whole 16-row decode step (4-bit weights once + 16 rows of " + ("8-bit (group 64) K / V)" if kv_bits else "bf16 K / V)"),
                        "achieved": step_bytes * steps_per_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": step_bytes * steps_per_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes_per_step": step_bytes, "weight_bytes_per_step": lm_params * 9 // 16,
                        "kv_bytes_per_step": n_req * ctx_mid * kv_tok}}
    return out


def _fast_w4_oracle_weights(W, predicate):
    """oracle weight dict with the accepted matrices as MLX 4-bit QW objects of RANDOM words / scales / biases (timing only:
    quantising 3.8 B weights on the host would take minutes; the values do not matter for a tokens/s sample)"""
    from oracle import quant as Q

    out = {}
    g = torch.Generator().manual_seed(7)
    for k, v in W.items():
        path = k[: -len(".weight")] if k.endswith(".weight") else None
        if path is not None and v.dim() == 2 and v.shape[1] % 64 == 0 and predicate(path, v):
            n, kk = v.shape
            wq = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, kk // 8), dtype=torch.int32, generator=g)
            sc = torch.full((n, kk // 64), 0.004, dtype=torch.bfloat16)
            bi = torch.full((n, kk // 64), -0.03, dtype=torch.bfloat16)
            out[k] = Q.QW(wq, sc, bi, 64, 4)
        else:
            out[k] = v
    return out


class ClockSampler:
    """GPU clocks while a config's child process runs (the Phi-3.5 4-bit line read 3450 vs 4200 tok/s in two
    contexts - clock / thermal state or the process?).  A thread samples the current shader and memory clock levels from the
    amdgpu sysfs tables (`pp_dpm_sclk` / `pp_dpm_mclk`: the line with the `*`), falling back to one `rocm-smi -c --json` call
    before and after when the tables are not there.  The parent never opens the device for this."""

    def __init__(self, period_s=0.25):
        import glob
        self.period = period_s
        self.paths = {}
        # which card is ours: a node's sysfs lists all eight, rocm-smi only the one this container was given
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        if cards:
            mine = [k for k in (self._smi_once() or {}).get("_cards", []) if os.path.isdir(f"/sys/class/drm/{k}/device")]
            for card in ([f"/sys/class/drm/{mine[0]}/device"] if mine else cards):
                if os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                    self.paths = {"sclk": os.path.join(card, "pp_dpm_sclk"), "mclk": os.path.join(card, "pp_dpm_mclk")}
                    break
        self.samples = {"sclk": [], "mclk": []}
        self.smi = []
        self._stop = None
        self._thread = None

    @staticmethod
    def _current_mhz(path):
        try:
            for ln in open(path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        except Exception:
            return None
        return None

    @staticmethod
    def _smi_once():
        import subprocess
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-c", "--json"], capture_output=True, text=True, timeout=15)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            out = {k.strip(" :"): v for k, v in card.items() if "sclk" in k or "mclk" in k}
            out["_cards"] = sorted(d)
            return out
        except Exception as e:
            return {"error": f"{type(e).__name__}: {e}"}

    def __enter__(self):
        import threading
        if not self.paths:
            self.smi.append(self._smi_once())
            return self
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                for k, pth in self.paths.items():
                    v = self._current_mhz(pth)
                    if v is not None:
                        self.samples[k].append(v)
                self._stop.wait(self.period)

        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2)
        else:
            self.smi.append(self._smi_once())
        return False

    def summary(self):
        if self._thread is None:
            return {"source": "rocm-smi -c --json, before / after", "before": self.smi[0] if self.smi else None,
                    "after": self.smi[1] if len(self.smi) > 1 else None}
        out = {"source": "%s, sampled every %.2f s" % (self.paths.get("sclk", ""), self.period)}
        for k, v in self.samples.items():
            if v:
                sv = sorted(v)
                out[k + "_mhz"] = {"min": sv[0], "median": sv[len(sv) // 2], "max": sv[-1], "n": len(sv)}
        return out


def other_configs(args, rank, ws, dev, t_start, budget_s=420.0):
    """The default line's `configs` block (only configs[1] had a driver-run line): a SHORT run of every
    other BASELINE config on this GPU - value, roofline and, where it fits in ~40 s, the oracle's CPU tokens/s - each in a
    try / except and under a wall-clock budget so that an extra can never cost the headline.  The full lines (more steps,
    vision rooflines, batched extras) are `--workload <name>`."""
    import gc
    import subprocess

    from mlx_vlm_amd.utils import cpu_quota

    plan = [("configs[0] nanollava", "nanollava", "nanollava", {}),
            ("configs[2] qwen2vl-7b-b32", "qwen2vl-7b-b32", "qwen2vl-7b", {}),
            ("configs[3] idefics2-b8", "idefics2-b8", "idefics2-8b", {}),
            ("configs[4] phi35v-w4-b16", "phi35v-w4-b16", "phi35v-w4", {}),
            ("configs[4] phi35v-w4-b16 kv_bits=8", "phi35v-w4-b16", None, {"kv_bits": 8})]
    keep_keys = ("metric", "value", "unit", "ms_per_step", "scaling", "dtype", "config", "roofline", "decode_tokens_per_s",
                 "decode_us_per_token", "e2e_tokens_per_s", "images_per_s_prefill", "prompt_tps", "kv_bits")
    block = {}
    for name, fn, kind, over in plan:
        if time.perf_counter() - t_start > budget_s:
            block[name] = {"skipped": f"wall-clock budget of the default line ({budget_s:.0f} s) reached"}
            continue
        # Each config runs as `bench.py --workload <name>` in a FRESH process (3 timed passes after 2 warm ones) and its JSON line
        # is read back: inside this process - after the headline model, the extras, the previous configs and the oracle's CPU
        # legs - the same workload read 17-20 % low (Phi-3.5: 3383-3491 vs 4182-4245 tok/s on its own, gpurun sessions 10 / 11 /
        # 12 and the first evidence run of round 4).  The CPU baselines stay here.
        t0 = time.perf_counter()
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", fn, "--steps", "3", "--warmup", "2", "--no-extras",
                   "--no-cpu-baseline"] + (["--kv-bits", str(over["kv_bits"])] if over.get("kv_bits") else [])
            with ClockSampler() as clk:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=max(60.0, budget_s - (time.perf_counter() - t_start) + 120.0))
            lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"rc={r.returncode}: {r.stderr.strip()[-300:]}")
            out = json.loads(lines[-1])
            row = {k: out[k] for k in keep_keys if k in out}
            row["gpu_wall_s"] = time.perf_counter() - t0
            row["gpu_clocks"] = clk.summary()
        except Exception as e:
            row = {"error": f"{type(e).__name__}: {e}"}
        gc.collect()
        torch.cuda.empty_cache()
        if kind and not args.no_cpu_baseline and "error" not in row and time.perf_counter() - t_start < budget_s:
            t1 = time.perf_counter()
            try:
                threads = min(cpu_quota(), 32)
                row["cpu_baseline"] = cpu_baseline_nanollava(threads) if kind == "nanollava" else cpu_baseline_lm(kind, threads, short=True)
                row["cpu_baseline"]["wall_s"] = time.perf_counter() - t1
            except Exception as e:
                row["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
            gc.collect()
        block[name] = row
    return block


def dry_run(args, rank, ws):
    """Everything of the N-rank job EXCEPT the kernels, on CPU ranks over gloo: the self-launch / torchrun environment, the
    rendezvous, the weight replica through parallel.WeightArena (rank 0 fills it, the others receive in place; contents
    verified on every rank), dp_batch_generate's length-sorted request deal + gather with a mock per-rank engine, and the
    timing protocol of the real line (warm-up, barrier, K steps, barrier, MAX over ranks).  The 8-GPU scaling run is the
    driver's; this keeps a launcher or collective typo from costing that run."""
    from mlx_vlm_amd import parallel

    dev = torch.device("cpu")
    t0 = time.perf_counter()
    W = {f"layers.{i}.w": (torch.full((1 << 20,), float(i + 1), dtype=torch.bfloat16) if rank == 0 else torch.empty(1 << 20, dtype=torch.bfloat16))
         for i in range(8)}
    W["table"] = torch.arange(4096, dtype=torch.int32) if rank == 0 else torch.empty(4096, dtype=torch.int32)
    parallel.barrier()
    t1 = time.perf_counter()
    parallel.broadcast_weights(W, src=0, bucket_bytes=4 << 20)
    bcast_s = parallel.max_over_ranks(time.perf_counter() - t1, dev)
    for i in range(8):
        assert float(W[f"layers.{i}.w"].float().mean()) == float(i + 1), ("broadcast", rank, i)
    assert W["table"][-1].item() == 4095
    nbytes = sum(v.numel() * v.element_size() for v in W.values())
    load = {"load_s": time.perf_counter() - t0, "weight_bytes": nbytes, "broadcast_s": bcast_s}
    max_tokens = args.max_tokens or 8

    def serve(indices, reqs, max_toks):                    # the per-rank engine: token j of request i = (i * 31 + j) % 997
        time.sleep(0.001 * len(indices))
        return [[(i * 31 + j) % 997 for j in range(max_toks[i])] for i in indices]

    rng = np.random.default_rng(0)                          # the same list on every rank
    reqs = [{"input_ids": np.arange(5 + int(rng.integers(0, 50)))} for _ in range(4 * ws + 1)]
    for _ in range(args.warmup):
        parallel.dp_batch_generate(None, None, requests=reqs, max_tokens=max_tokens, serve=serve)
    parallel.barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = parallel.dp_batch_generate(None, None, requests=reqs, max_tokens=max_tokens, serve=serve)
    parallel.barrier()
    my_wall = time.perf_counter() - t0
    wall = parallel.max_over_ranks(my_wall, dev)
    # the same per-rank vectors the real line carries (stage_headline), through the same collective
    decode_ranks = parallel.per_rank(my_wall, dev)
    thread_ranks = parallel.per_rank(float(torch.get_num_threads()), dev)
    if rank != 0:
        return None
    assert res["tokens"] == [[(i * 31 + j) % 997 for j in range(max_tokens)] for i in range(len(reqs))]
    info = _dist_info(ws, load)
    return {"metric": "decode tokens/sec + vision-prefill images/sec, Qwen2-VL-2B", "dry_run": True,
            "value": args.steps * res["generation_tokens"] / wall, "unit": "tokens/s", "n_gpus": ws, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "none (no kernels)", "data": "synthetic",
            "config": {"workload": "DRY RUN: mock per-rank engine over gloo CPU ranks - launcher / rendezvous / weight arena "
                                   "broadcast / request deal / timing protocol only", "parallelism": f"dp{ws}"},
            "load": load, "distributed": info, "per_rank_requests": res["per_rank_requests"],
            "host_prep_s_per_rank": res["host_prep_s_per_rank"], "serve_s_per_rank": res["serve_s_per_rank"],
            "decode_s_per_rank": decode_ranks, "torch_threads_per_rank": [int(v) for v in thread_ranks]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--max-tokens", type=int, default=0, help="0 = the workload's own (256 / 64 / 64)")
    ap.add_argument("--lookahead", type=int, default=8)
    ap.add_argument("--vit-batch", type=int, default=64,
                    help="336x336 images per vision-tower call of the ViT throughput line (16 = the workload of rounds 1-4 is reported "
                         "next to it; 68 images, where every GEMM fills its last round of 256 workgroups, measured +0.4 %: the tail of a "
                         "launch is not a whole round)")
    ap.add_argument("--workload", default="qwen2vl-2b", choices=["qwen2vl-2b", "nanollava", "qwen2vl-7b-b32", "qwen2vl-2b-w4", "phi35v-w4-b16", "idefics2-b8"])
    ap.add_argument("--kv-bits", type=int, default=0, help="phi35v-w4-b16: 8 = uniform 8-bit KV cache (kv_bits of the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-hf", action="store_true", help="skip the HuggingFace torch-CPU second opinion of cpu_baseline")
    ap.add_argument("--no-extras", action="store_true", help="skip kernel rooflines / ViT throughput (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of the other BASELINE configs in the default line")
    ap.add_argument("--stage", default="", choices=["", "headline", "extras"],
                    help="internal: one stage of the default single-GPU line in a process of its own (see orchestrate)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank path: launcher, rendezvous (gloo), weight-arena broadcast, request deal, "
                         "timing protocol, result gather, JSON line - no kernels (tests/test_host_cpu.py runs it with --gpus 2)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` outside torchrun: launch the N ranks ourselves (one process per GPU over RCCL, the form
        # the driver uses: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        print(f"[bench] --gpus {args.gpus} outside torchrun: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
        sys.exit(subprocess.call(cmd, env=env))
    ws_env = int(os.environ.get("WORLD_SIZE", "1"))
    if ws_env != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws_env}: the rank count of the launcher and --gpus must agree "
                 "(run `python bench.py --gpus N`, or torchrun --nproc-per-node N bench.py --gpus N)")

    from mlx_vlm_amd import parallel, synthetic

    if args.dry_run:
        rank, ws, local = parallel.init(backend="gloo")
        out = dry_run(args, rank, ws)
        if rank == 0:
            print(json.dumps(out), flush=True)
        parallel.barrier()
        parallel.shutdown()
        return
    ws = ws_env
    if args.workload != "qwen2vl-2b":
        rank, ws, local = parallel.init()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if rank == 0 and ws > 1:
            print(f"[bench] {ws} ranks, backend {_dist_info(ws)['backend']} (RCCL over xGMI), one process per GPU", file=sys.stderr, flush=True)
        out = {"nanollava": workload_nanollava, "qwen2vl-7b-b32": workload_7b_b32, "qwen2vl-2b-w4": workload_2b_w4,
               "phi35v-w4-b16": workload_phi35v_w4_b16, "idefics2-b8": workload_idefics2_b8}[args.workload](
            args, rank, ws, dev)
        kind = {"qwen2vl-7b-b32": "qwen2vl-7b", "qwen2vl-2b-w4": "qwen2vl-2b-w4", "phi35v-w4-b16": "phi35v-w4",
                "idefics2-b8": "idefics2-8b"}.get(args.workload)
        if kind:
            out = _with_cpu_baseline(out, kind, args, rank, ws)
        if rank == 0:
            print(json.dumps(out), flush=True)
        parallel.barrier()
        parallel.shutdown()
        return
    args.max_tokens = args.max_tokens or 256
    if args.stage == "headline" or ws > 1:
        # one rank of the N-rank job (torchrun / the self-launch above), or the headline child of the single-GPU orchestrator
        rank, ws, local = parallel.init()
        out = stage_headline(args, rank, ws, local)
        if rank == 0:
            if not args.stage:
                out.pop("_traffic_gate_up", None)          # (the orchestrator's hand-over field)
            print(json.dumps(out), flush=True)
        parallel.barrier()
        parallel.shutdown()
        return
    if args.stage == "extras":
        print(json.dumps(stage_extras(args)), flush=True)
        return
    orchestrate(args)


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


def stage_extras(args):
    """Everything of the default line that is not the headline's timed region, in a process of its own (a GPU memory fault
    cannot be caught by try / except: BENCH_r04): per-kernel rooflines, ViT throughput, batched / continuous / sampled decode."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.models import qwen2_vl

    rank, ws, local = parallel.init()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg, model, load = _load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, rank, dev, kv_pool_tokens=32768, max_seqs=40)
    req = build_request(cfg, 448, 128, seed=rank)
    req = (req[0], req[1].to(dev), req[2])
    run_step(model, req, 16, args.lookahead)
    extras = {}

    def emit():          # the parent reads the LAST complete line: every finished extra survives a later fault
        print(json.dumps(extras), flush=True)

    extras["kernels"] = kernel_rooflines(model, cfg)
    emit()
    ips336, dt336 = vit_throughput(model, cfg, args.vit_batch, 336)
    ips448, dt448 = vit_throughput(model, cfg, 1, 448)
    extras.update(vit336=(ips336, dt336), vit448=(ips448, dt448))
    emit()
    sweep = {}
    for nb in (16, 32, 64, 128, 136):                 # the same tower at other batch sizes (16 = the workload of rounds 1-4;
                                                      # 136 x 576 patches = 306 row tiles of 256: whole rounds of the 256 CUs)
        if nb != args.vit_batch:
            ips, dt = vit_throughput(model, cfg, nb, 336)
            sweep[str(nb)] = {"images_per_s": ips, "ms_per_call": dt * 1e3, "frac_of_mfma_peak": ips * VIT_TFLOP_336 / MFMA_BF16_PEAK_TF}
    extras["vit336_sweep"] = sweep
    emit()
    for key, fn in (("batch8", lambda: batch_decode_throughput(model, cfg, 8, 64)),
                    ("batch16", lambda: batch_decode_throughput(model, cfg, 16, 64)),
                    ("wide64", lambda: wide_decode_throughput(model, cfg, 64, 48)),
                    ("continuous", lambda: continuous_batch_throughput(model, cfg)),
                    ("sampled", lambda: sampled_decode_throughput(model, req, 128, args.lookahead))):
        try:
            extras[key] = fn()
        except Exception as e:   # an extra must never cost the line
            extras[key] = {"error": f"{type(e).__name__}: {e}"}
        emit()
    return extras


def _child(args, stage, timeout_s):
    """Run one stage of the default line as `bench.py --stage <stage>` -> (last JSON line or None, returncode, stderr tail)."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--stage", stage, "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--max-tokens", str(args.max_tokens), "--lookahead", str(args.lookahead), "--vit-batch", str(args.vit_batch)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        rc, so, se = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        def _txt(x):
            return x.decode(errors="replace") if isinstance(x, bytes) else (x or "")
        rc, so, se = -9, _txt(e.stdout), _txt(e.stderr) + f"\n[bench] stage {stage}: timeout after {timeout_s} s"
    lines = [ln for ln in so.strip().splitlines() if ln.startswith("{")]
    doc = None
    for ln in reversed(lines):
        try:
            doc = json.loads(ln)
            break
        except ValueError:          # a line cut short by an abort
            continue
    return doc, rc, se.strip()[-600:]


def orchestrate(args):
    """`python bench.py` on one GPU: this process never touches the GPU.  The headline (timed region) runs in a child whose JSON
    is echoed to stderr the moment it exists; the extras, the CPU baseline and the other configs follow, each in a process of
    its own, and ONE enriched line goes to stdout at the end.  A headline child killed by a signal is re-run (up to 3 attempts,
    every failed attempt is reported in `headline_attempts`)."""
    attempts = []
    out = None
    for _ in range(3):
        with ClockSampler() as clk:
            out, rc, err = _child(args, "headline", 900)
        if out is not None and rc == 0:
            break
        attempts.append({"rc": rc, "stderr_tail": err[-300:]})
        print(f"[bench] headline attempt {len(attempts)} failed (rc {rc}): {err[-300:]}", file=sys.stderr, flush=True)
        out = None
    if out is None:
        sys.exit(f"bench.py: the headline stage failed {len(attempts)} times: {attempts}")
    if attempts:
        out["headline_attempts"] = {"failed": attempts, "succeeded_on": len(attempts) + 1}
    # in `config` (which the driver's record keeps): a re-run headline must be impossible to miss
    out["config"]["headline_retries"] = len(attempts)
    out["config"]["vit_batch"] = args.vit_batch
    traffic_gu = out.pop("_traffic_gate_up", None)
    out["gpu_clocks"] = clk.summary()
    print("[bench] headline: " + json.dumps(out), file=sys.stderr, flush=True)

    extras = None
    if not args.no_extras:
        extras, rc, err = _child(args, "extras", 900)
        if rc != 0:
            extras = dict(extras or {})
            extras["error"] = f"extras stage rc {rc}: {err[-300:]}"
            print(f"[bench] extras stage failed (rc {rc}): {err[-300:]}", file=sys.stderr, flush=True)
    if extras:
        if extras.get("kernels"):
            k = extras["kernels"]["gemv_gate_up_swiglu"]
            out["roofline_kernel"] = {"bound": "hbm", "kernel": GATE_UP_KERNEL + " (RMSNorm + gate/up GEMV + SwiGLU, 28 launches/token)",
                                      "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["GBps"] / HBM_PEAK_GBS,
                                      "traffic": traffic_gu, "bytes_per_launch": k["bytes_per_launch"], "us_per_launch": k["us_per_launch"]}
            out["kernel_rooflines"] = extras["kernels"]
        if extras.get("vit336"):
            ips336, dt336 = extras["vit336"]
            ips448, dt448 = extras["vit448"]
            out["vision_images_per_s"] = ips336
            out["roofline_vit"] = {"bound": "mfma", "achieved": ips336 * VIT_TFLOP_336, "peak": MFMA_BF16_PEAK_TF,
                                   "unit": "TFLOP/s", "frac": ips336 * VIT_TFLOP_336 / MFMA_BF16_PEAK_TF, "traffic": None,
                                   "workload": f"{args.vit_batch} x 336x336 images per call ({args.vit_batch * 576} patches)",
                                   "ms_per_call": dt336 * 1e3}
            out["vision_single_448_images_per_s"] = ips448
            out["vision_single_448_tflops"] = ips448 * VIT_TFLOP_448
            out["vision_batch_sweep_336"] = extras.get("vit336_sweep")
            # the images/s half of BASELINE's metric INSIDE `roofline` (the object the driver's record keeps whole), like for like:
            # the quoted batch, the 16-image workload of rounds 1-4, and the single 448 x 448 image of configs[1] - all against
            # the same 2.5 PF (reference definition of the timed call: mlx_vlm/models/qwen2_vl/vision.py:257-290)
            def _vit(ips, ms, batch, tflop):
                return {"images_per_s": ips, "frac": ips * tflop / MFMA_BF16_PEAK_TF, "batch": batch, "ms_per_call": ms,
                        "achieved_tflops": ips * tflop, "peak_tflops": MFMA_BF16_PEAK_TF}
            out["roofline"]["vit"] = _vit(ips336, dt336 * 1e3, args.vit_batch, VIT_TFLOP_336)
            s16 = (extras.get("vit336_sweep") or {}).get("16")
            if args.vit_batch == 16:
                out["roofline"]["vit_16"] = dict(out["roofline"]["vit"])
            elif s16:
                out["roofline"]["vit_16"] = _vit(s16["images_per_s"], s16["ms_per_call"], 16, VIT_TFLOP_336)
            out["roofline"]["vit_single_448"] = _vit(ips448, dt448 * 1e3, 1, VIT_TFLOP_448)
            # what the matrix cores sustain on random operands with nothing else running (profiles/r06_gemm_power_limit.txt:
            # the chip is power-limited there - 1.87 PF at 1.85 GHz for v_mfma_f32_16x16x32_bf16, 2.37 PF on zeros)
            out["roofline"]["vit"]["frac_of_random_operand_mfma_ceiling"] = ips336 * VIT_TFLOP_336 / MFMA_RANDOM_OPERAND_TF
        for src, dst in (("batch8", "batch8_decode"), ("batch16", "batch16_decode"), ("wide64", "wide64_decode"),
                         ("continuous", "continuous_batching"), ("sampled", "sampled_decode")):
            out[dst] = extras.get(src)
        if extras.get("error"):
            out["extras_error"] = extras["error"]
    if not args.no_cpu_baseline:
        try:
            from mlx_vlm_amd.utils import cpu_quota
            out["cpu_baseline"] = cpu_baseline(min(cpu_quota(), 32), with_hf=not args.no_cpu_hf)          # the cores the container may use
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_extras and not args.no_configs:
        try:
            out["configs"] = other_configs(args, 0, 1, None, T_PROCESS_START)
        except Exception as e:
            out["configs"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
