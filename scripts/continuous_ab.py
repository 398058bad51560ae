"""A/B of the BatchGenerator admission policies inside ONE process (box-to-box variance is 5-10 %):
synchronous admission, asynchronous (side stream) without / with prefill-ahead, and static batches of 8.
Usage: python scripts/continuous_ab.py [n_requests] [--breakdown] [--gc-log] [--no-gc-freeze]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mlx_vlm_amd.batch import BatchGenerator  # noqa: E402
from mlx_vlm_amd.generate import batch_generate_ids  # noqa: E402
from mlx_vlm_amd import synthetic  # noqa: E402
from mlx_vlm_amd.models.qwen2_vl import Model, ModelConfig  # noqa: E402


def watch_gc():
    """Print every cyclic-GC pass of generation >= 1 with its duration: a 100 ms generation-2 pass over the torch /
    transformers heap is the prime suspect for the sporadic host stalls inside prefill enqueues."""
    import gc

    t0 = {}

    def cb(phase, info):
        if phase == "start":
            t0[info["generation"]] = time.perf_counter()
        elif info["generation"] >= 1:
            print(f"    [gc] generation {info['generation']} pass: {1e3 * (time.perf_counter() - t0.get(info['generation'], 0)):.1f} ms, "
                  f"collected {info['collected']}")

    gc.callbacks.append(cb)


def main():
    if "--gc-log" in sys.argv:
        watch_gc()
    n_requests = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 24
    cfg = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))
    W = synthetic.random_weights(cfg, seed=0, device="cuda", fill=True)
    model = Model(cfg, device="cuda", kv_pool_tokens=16384, max_seqs=16)
    model.load_weights(W)
    del W
    if "--no-gc-freeze" not in sys.argv:
        from mlx_vlm_amd.utils import freeze_heap
        freeze_heap()
    reqs = [bench.build_request(cfg, 336, 64, 700 + i) for i in range(n_requests)]
    ids = [r[0].reshape(-1) for r in reqs]
    pix = [r[1] for r in reqs]
    thw = [r[2] for r in reqs]
    lens = [24 + (37 * i) % 73 for i in range(n_requests)]

    def run(**kw):
        gen = BatchGenerator(model, None, completion_batch_size=8, prefill_batch_size=8, compute_logprobs=False, **kw)
        gen.insert(ids, lens, prompt_kwargs=[dict(pixel_values=p, image_grid_thw=g) for p, g in zip(pix, thw)])
        n = rounds = 0
        while gen.has_work:
            n += len(gen.next()[1])
            rounds += 1
        steps = gen._steps_counter
        gen.close()
        return n, rounds, steps

    def static():
        n = 0
        for i in range(0, n_requests, 8):
            sl = slice(i, i + 8)
            toks, _ = batch_generate_ids(model, ids[sl], pix[sl], thw[sl], max_tokens=max(lens[sl]))
            n += sum(min(len(t), m) for t, m in zip(toks, lens[sl]))
        return n, 0, 0

    if "--breakdown" in sys.argv:
        from mlx_vlm_amd import ops
        from mlx_vlm_amd.generate import embed_requests

        def T():
            torch.cuda.synchronize()
            return time.perf_counter()

        # (a) steady decode rounds, 8 rows, no admissions
        for logp in (False, True):
            gen = BatchGenerator(model, None, completion_batch_size=8, prefill_batch_size=8, compute_logprobs=logp)
            gen.insert(ids[:8], [400] * 8, prompt_kwargs=[dict(pixel_values=p, image_grid_thw=g) for p, g in zip(pix[:8], thw[:8])])
            for _ in range(20):
                gen.next()
            t0 = T()
            host = 0.0
            for _ in range(100):
                h0 = time.perf_counter()
                gen.next()
                host += time.perf_counter() - h0
            t1 = T()
            print(f"steady 8 rows logprobs={logp}: {(t1 - t0) * 10:.3f} ms/round, host time inside next() {host * 10:.3f} ms/round")
            gen.close()
        # (b) one admission, phase by phase (main stream, synchronised between phases)
        lm = model.language_model
        for rep in range(3):
            for k in (1, 8):
                t0 = T()
                emb, pos, ln, deltas = embed_requests(model, ids[:k], pix[:k], thw[:k])
                h1 = time.perf_counter()
                t1 = T()
                caches = [lm.make_cache() for _ in range(k)]
                for c, L in zip(caches, ln):
                    c[0]._seq.reserve(L + 66)
                t2 = T()
                logits = lm.prefill(emb, pos, caches, ln, "last")
                h3 = time.perf_counter()
                t3 = T()
                tok0, _ = ops.sample(logits, step=torch.zeros(1, dtype=torch.int32, device="cuda"), want_logprobs=False)
                t4 = T()
                for c in caches:
                    c[0]._seq.release()
                print(f"rep{rep} admission k={k}: embed+ViT {1e3 * (t1 - t0):.2f} ms (host enqueue {1e3 * (h1 - t0):.2f}), "
                      f"cache+reserve {1e3 * (t2 - t1):.2f}, prefill {1e3 * (t3 - t2):.2f} (host enqueue {1e3 * (h3 - t2):.2f}), "
                      f"sample {1e3 * (t4 - t3):.2f}")
        return

    variants = [("sync", lambda: run(async_prefill=False)), ("async_ahead0", lambda: run(prefill_ahead=0)),
                ("async_ahead2", lambda: run(prefill_ahead=2)), ("async_ahead4", lambda: run(prefill_ahead=4)),
                ("static8", static)]
    for name, fn in variants:
        fn()
    for rep in range(3):
        for name, fn in variants:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n, rounds, steps = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"rep{rep} {name:13s} {n / dt:8.1f} useful tok/s  ({n} tokens, {dt * 1e3:6.1f} ms, {rounds} rounds, {steps} steps)")


if __name__ == "__main__":
    main()
