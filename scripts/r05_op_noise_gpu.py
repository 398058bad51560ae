#!/usr/bin/env python
"""Which kernel contributes most to the distance between the engine and the oracle (round-4 review, item 5)?  Every op of a
Qwen2-VL-2B decoder layer on the SAME bf16 inputs, three ways: the HIP kernel, the oracle's op (fp32 arithmetic, one rounding -
its statement of MLX) and the exactly rounded result (float64 arithmetic, one rounding to bf16).  rel-rms of the output against
the exact one.  An op whose HIP distance is the oracle's distance adds nothing of its own (two fp32 summation orders, both one
rounding away from exact); an op that is clearly above it has a rounding point the typed graph does not have.
usage (GPU): python scripts/r05_op_noise_gpu.py [out.txt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ops as O            # (a script under scripts/: the oracle is the checker here, as in tests/)
from mlx_vlm_amd import ops as vops

BF, F64 = torch.bfloat16, torch.float64
torch.set_num_threads(min(16, os.cpu_count() or 8))
g = torch.Generator().manual_seed(5)


def rnd(*shape, s=1.0):
    return (torch.randn(*shape, generator=g) * s).to(BF)


def rr(a, b):
    a, b = a.detach().cpu().to(F64), b.detach().cpu().to(F64)
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


lines = ["op (2B decoder-layer shapes, T tokens)                     HIP vs exact   oracle vs exact   HIP vs oracle"]


def row(name, hip, orc, exact):
    lines.append(f"{name:58s} {rr(hip, exact):.3e}      {rr(orc, exact):.3e}        {rr(hip, orc):.3e}")
    print(lines[-1], flush=True)


T, D, I, Hq, Hkv, hd = 64, 1536, 8960, 12, 2, 128
x = rnd(T, D)
# RMSNorm (typed: bf16(x * inv) * w -> bf16)
w_n = (1 + 0.1 * torch.randn(D, generator=g)).to(BF)
xf = x.to(F64)
inv = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
exact = ((xf * inv).to(BF).to(F64) * w_n.to(F64)).to(BF)
row("RMSNorm", vops.rmsnorm(x.cuda(), w_n.cuda()), O.rms_norm(x, w_n, 1e-6), exact)
# qkv projection with bias
W = rnd(2048, D, s=0.02); b = rnd(2048, s=0.1)
exact = (x.to(F64) @ W.to(F64).T + b.to(F64)).to(BF)
row("qkv GEMM + bias (K = 1536)", vops.gemm(x.cuda(), W.cuda(), bias=b.cuda(), epilogue=vops.EPI_BIAS), O.linear(x, W, b), exact)
# gate/up + SwiGLU (typed: gate, up -> bf16; silu(gate) * up with the oracle's rounding points)
Wgu = rnd(2 * I, D, s=0.02)
gu = (x.to(F64) @ Wgu.to(F64).T).to(BF)
# the product interleaves gate / up rows; build both layouts from the same matrices
Wg, Wu = Wgu[:I], Wgu[I:]
Wil = torch.stack([Wg, Wu], dim=1).reshape(2 * I, D).contiguous()
exact = O.swiglu((x.to(F64) @ Wg.to(F64).T).to(BF), (x.to(F64) @ Wu.to(F64).T).to(BF))
orc = O.swiglu(O.linear(x, Wg), O.linear(x, Wu))
row("gate/up GEMM + SwiGLU (K = 1536)", vops.gemm(x.cuda(), Wil.cuda(), epilogue=vops.EPI_SWIGLU), orc, exact)
# down projection + residual (K = 8960)
a = rnd(T, I, s=0.3); Wd = rnd(D, I, s=0.02); res = rnd(T, D)
exact = ((a.to(F64) @ Wd.to(F64).T).to(BF).to(F64) + res.to(F64)).to(BF)
row("down GEMM + residual (K = 8960)", vops.gemm(a.cuda(), Wd.cuda(), res=res.cuda(), epilogue=vops.EPI_RESIDUAL), O.add(res, O.linear(a, Wd)), exact)
# head row (one token, V = 151,936): the decode GEMV
Wh = rnd(151936, D, s=0.02); x1 = rnd(1, D)
exact = (x1.to(F64) @ Wh.to(F64).T).to(BF)
row("lm_head GEMV (one row, V = 151,936)", vops.gemv(x1.cuda(), Wh.cuda()), O.linear(x1, Wh), exact)
# causal flash attention, GQA 12:2, head 128
for Tn in (64, 384):
    q, k, v = rnd(Tn, Hq, hd), rnd(Tn, Hkv, hd), rnd(Tn, Hkv, hd)
    qo, ko, vo = (t.permute(1, 0, 2)[None] for t in (q, k, v))
    orc = O.sdpa(qo, ko, vo, hd ** -0.5, causal=True)[0].permute(1, 0, 2)
    rep = Hq // Hkv
    s = (qo.to(F64) @ ko.to(F64).repeat_interleave(rep, 1).transpose(-1, -2)) * hd ** -0.5
    i, j = torch.arange(Tn)[:, None], torch.arange(Tn)[None, :]
    s = s.masked_fill(~(j <= i), float("-inf"))
    exact = (torch.softmax(s, -1) @ vo.to(F64).repeat_interleave(rep, 1)).to(BF)[0].permute(1, 0, 2)
    qkv = torch.cat([q.reshape(Tn, -1), k.reshape(Tn, -1), v.reshape(Tn, -1)], 1).cuda()
    cu = torch.tensor([0, Tn], dtype=torch.int32).cuda()
    out = vops.attn_prefill(qkv, qkv[:, Hq * hd:], qkv[:, (Hq + Hkv) * hd:], cu, (Tn + 127) // 128, Hq, Hkv, hd, hd ** -0.5, True)
    row(f"causal flash attention, {Tn} tokens (P in bf16 for P.V)", out.view(Tn, Hq, hd), orc, exact)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(__doc__ + "\n" + "\n".join(lines) + "\n")
