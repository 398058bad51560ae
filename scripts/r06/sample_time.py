"""us per vops.sample call (logits in, V = 151936, one row) per filter set; VLM_SAMPLE_SPLIT=0 in the env = the one-workgroup kernel."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mlx_vlm_amd import ops
g = torch.Generator().manual_seed(1)
logits = (torch.randn(1, 151936, generator=g) * 2.5 + 3.0).to(torch.bfloat16).cuda()
st = torch.zeros(1, dtype=torch.int32, device="cuda")
ws = ops.sample_workspace(1, "cuda")
out = {}
for name, kw in (("temp", {}), ("top_p", dict(top_p=0.9)), ("min_p", dict(min_p=0.02)), ("top_k", dict(top_k=50)),
                 ("top_p_min_p", dict(top_p=0.9, min_p=0.02)), ("chain", dict(top_p=0.9, min_p=0.02, top_k=50))):
    f = lambda: ops.sample(logits, temperature=0.7, seed=3, step=st, want_logprobs=True, ws=ws, **kw)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(20): f()
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) * 1e3 / 200, 2)
print(json.dumps(out))
