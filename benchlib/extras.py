"""Everything of the default line that is not the headline's timed region (its own process: bench.py --stage extras): per-kernel
rooflines, ViT throughput, batched / wide / continuous / sampled decode."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch
from .common import *  # noqa: F401,F403

__all__ = ['sampled_decode_throughput', 'kernel_rooflines', 'vit_throughput', 'batch_decode_throughput', 'wide_decode_throughput', 'continuous_batch_throughput', 'stage_extras']


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
