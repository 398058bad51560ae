"""The `cpu_baseline` legs of the bench lines: the oracle (tests' checker, allowed here as the CPU baseline ONLY) timed on the host
cores on a bounded sample of each workload, plus the HuggingFace torch-CPU second opinion."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch
from .common import *  # noqa: F401,F403

__all__ = ['_fast_w4_oracle_weights', 'cpu_baseline', '_hf_cpu_baseline', 'cpu_baseline_nanollava', 'cpu_baseline_lm', '_with_cpu_baseline']


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
