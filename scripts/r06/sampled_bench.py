"""Sampled decode (temperature / top-p / the chain) on the bench's 2B request, with a greedy pass beside it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from benchlib.common import *      # noqa
from benchlib.workloads import *   # noqa
from benchlib.extras import sampled_decode_throughput
from benchlib.headline import *    # noqa
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl

dev = torch.device("cuda", 0)
cfg, model, load = _load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=32768, max_seqs=40)
req = build_request(cfg, 448, 128, seed=0)
req = (req[0], req[1].to(dev), req[2])
run_step(model, req, 16, 8)
out = sampled_decode_throughput(model, req, 128, 8)
from mlx_vlm_amd.generate import generate_step
for rep in range(2):
    n, t0 = 0, None
    for _ in generate_step(req[0], model, req[1], None, max_tokens=128, temperature=0.0, image_grid_thw=req[2], return_logprobs=False, lookahead=8):
        if t0 is None:
            t0 = time.perf_counter()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
out["greedy"] = {"generation_tps": (n - 1) / dt}
print(json.dumps(out))
