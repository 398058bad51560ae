"""One sampled decode of the bench's 2B request with the filter set named in argv[1] (temp | top_p | chain) - for rocprofv3."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from benchlib.common import *      # noqa
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl
from mlx_vlm_amd.generate import generate_step

kw = {"temp": {}, "top_p": dict(top_p=0.9), "chain": dict(top_p=0.9, min_p=0.02, top_k=50)}[sys.argv[1]]
dev = torch.device("cuda", 0)
cfg, model, load = _load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=32768, max_seqs=40)
req = build_request(cfg, 448, 128, seed=0)
req = (req[0], req[1].to(dev), req[2])
for rep in range(2):
    n, t0 = 0, None
    for _ in generate_step(req[0], model, req[1], None, max_tokens=128, temperature=0.7, seed=1234, image_grid_thw=req[2], return_logprobs=False, lookahead=8, **kw):
        if t0 is None:
            t0 = time.perf_counter()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(json.dumps({sys.argv[1]: (n - 1) / dt}))
