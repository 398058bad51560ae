#!/usr/bin/env python
"""Decode-kernel micro-benchmarks on the live 2B weights (HIP events): cold (cycle all 28 layers, 1.5 GB) vs
warm (same layer repeatedly: fits the 256 MB Infinity Cache) - tells how much an HBM->MALL prefetch could buy."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_vlm_amd import ops, synthetic
from mlx_vlm_amd.models.qwen2_vl import Model, ModelConfig

def ev(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

cfg = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))
W = synthetic.random_weights(cfg, seed=0, device="cuda")
model = Model(cfg, kv_pool_tokens=8192, max_seqs=8); model.load_weights(W); del W
lm = model.language_model; t = cfg.text_config
D, I, V, L = t.hidden_size, t.intermediate_size, t.vocab_size, t.num_hidden_layers
bf = torch.bfloat16
x = torch.randn(1, D, device="cuda").to(bf); act = torch.randn(1, I, device="cuda").to(bf)
o_gu = torch.empty(1, I, dtype=bf, device="cuda"); h = torch.zeros(1, D, dtype=bf, device="cuda")
qkv = torch.empty(1, 2048, dtype=bf, device="cuda")
res = {}
def cold(f): return ev(lambda: [f(i) for i in range(L)], 3) / L
def warm(f): return ev(lambda: [f(0) for _ in range(L)], 3) / L
gu = lambda i: ops.gemv(x, lm._w[f"{i}.wgu"], norm_w=lm._w[f"{i}.ln2"], out=o_gu, epilogue=ops.EPI_SWIGLU)
dn = lambda i: ops.gemv(act, lm._w[f"{i}.wdown"], res=h, out=h, epilogue=ops.EPI_RESIDUAL)
op = lambda i: ops.gemv(x, lm._w[f"{i}.wo"], res=h, out=h, epilogue=ops.EPI_RESIDUAL)
qk = lambda i: ops.gemv(x, lm._w[f"{i}.wqkv"], bias=lm._w[f"{i}.bqkv"], norm_w=lm._w[f"{i}.ln1"], out=qkv, epilogue=ops.EPI_BIAS)
for name, f, nb in (("gate_up", gu, 4*I*D), ("down", dn, 2*I*D), ("o_proj", op, 2*D*D), ("qkv", qk, 2*2048*D)):
    c, w = cold(f), warm(f)
    res[name] = dict(cold_us=c, warm_us=w, cold_GBps=nb/c/1e3, warm_GBps=nb/w/1e3, MB=nb/1e6)
# empty-ish kernel launch floor: tiny gemv (N=64) back to back
wt = torch.randn(64, D, device="cuda").to(bf); ot = torch.empty(1, 64, dtype=bf, device="cuda")
res["tiny_gemv_us"] = ev(lambda: [ops.gemv(x, wt, out=ot) for _ in range(50)], 3) / 50
# plain copy bandwidth (torch) as the achievable-HBM yardstick
big = torch.empty(1 << 29, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(big)
us = ev(lambda: dst.copy_(big), 5); res["copy_GBps_rw"] = 2 * big.numel() / us / 1e3
print(json.dumps(res, indent=1))
