"""an 8k-token prompt (text only, 2B dims) through generate_step: chunked prefill (prefill_step_size 2048, the default)
against one-shot (prefill_step_size None) - time to the first token and the first 8 tokens of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from benchlib.common import _load_synthetic
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl
from mlx_vlm_amd.generate import generate_step

dev = torch.device("cuda", 0)
cfg, model, load = _load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=65536, max_seqs=8)
for L in (4096, 8192, 16384):
    ids = torch.from_numpy(np.random.default_rng(L).integers(1000, 100000, (1, L)))
    for name, step in (("chunked 2048", 2048), ("one-shot", None)):
        best, toks = 1e9, None
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = []
            for i, (tok, _) in enumerate(generate_step(ids, model, None, None, max_tokens=8, temperature=0.0, prefill_step_size=step, return_logprobs=False)):
                if i == 0:
                    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                out.append(int(tok))
            toks = out
        print(f"L={L:6d} {name:13s} first token {best*1e3:8.2f} ms  ({L/best:9.0f} prompt tok/s)  tokens {toks}")
