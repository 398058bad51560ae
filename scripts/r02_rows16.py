"""Continuous batching at 8 vs 16 decode rows (32 requests, 336x336 image + 64 text tokens, 64 new tokens each)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.batch import BatchGenerator
from mlx_vlm_amd.models import qwen2_vl
dev = torch.device("cuda", 0)
cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=32768, max_seqs=40)
n = 32
reqs = [bench.build_request(cfg, 336, 64, 700 + i) for i in range(n)]
ids = [r[0].reshape(-1) for r in reqs]
kw = [dict(pixel_values=r[1], image_grid_thw=r[2]) for r in reqs]
for rows in (8, 16, 8, 16):
    for rep in range(2):
        gen = BatchGenerator(model, None, completion_batch_size=rows, prefill_batch_size=rows, compute_logprobs=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gen.insert(ids, [64] * n, prompt_kwargs=kw)
        tot = 0
        while gen.has_work:
            tot += len(gen.next()[1])
        gen.close(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"rows={rows}: {tot / dt:.0f} tok/s end to end ({tot} tokens, {dt * 1e3:.0f} ms, steps {gen._steps_counter})", flush=True)
