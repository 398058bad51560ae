"""The other BASELINE.json configs as workloads of their own (bench.py --workload ...) and their short runs inside the default line."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch
from .common import *  # noqa: F401,F403
from .cpu_baseline import *  # noqa: F401,F403
from .extras import *  # noqa: F401,F403

__all__ = ['workload_nanollava', 'workload_2b_w4', 'workload_7b_b32', 'workload_idefics2_b8', 'workload_phi35v_w4_b16', 'other_configs']


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
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", fn, "--steps", "3", "--warmup", "2", "--no-extras",
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
