#!/usr/bin/env python
"""Batched decode run for profiling: B requests (336^2 image + 128 text tokens each) through batch_generate_ids."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.generate import batch_generate_ids
from mlx_vlm_amd.models.qwen2_vl import Model, ModelConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ntok = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))
W = synthetic.random_weights(cfg, seed=0, device="cuda")
model = Model(cfg, kv_pool_tokens=16384, max_seqs=16); model.load_weights(W); del W
reqs = [bench.build_request(cfg, 336, 128, 500 + i) for i in range(B)]
ids, pix, thw = [r[0].reshape(-1) for r in reqs], [r[1] for r in reqs], [r[2] for r in reqs]
batch_generate_ids(model, ids, pix, thw, max_tokens=8)
torch.cuda.synchronize(); t0 = time.perf_counter()
toks, stats = batch_generate_ids(model, ids, pix, thw, max_tokens=ntok)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"B={B}: generation {stats.generation_tps:.0f} tok/s, prompt {stats.prompt_tps:.0f} tok/s, e2e {sum(len(t) for t in toks)/dt:.0f} tok/s, "
      f"decode step {1e3 * B / max(stats.generation_tps, 1e-9):.3f} ms")
