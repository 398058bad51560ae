"""Per-kernel distance from the EXACTLY ROUNDED result: every op of a Qwen2-VL-2B decoder layer on
the same bf16 inputs - the HIP kernel against float64 arithmetic rounded ONCE to bf16 at the reference's rounding points, the
oracle's fp32 op beside it.  rel-rms of the bf16 outputs: an op that only sums in another fp32 order differs from the exact
result on a few elements per thousand (1e-5 .. 1e-4); an op with a rounding point of its own sits at ~1e-3 (every element off
by a fraction of an ulp).  The bars catch a regression of ANY kernel without leaning on the 4e-2 full-depth bar:

    GEMM / GEMV / norms                      <= 1e-4
    decode attention (P as hi + lo bf16)     <= 5e-4     (one bf16 P operand: 1.8e-3, profiles/r05_engine_noise_by_depth_and_op.txt)
    prompt flash attention, D = 128 (hi + lo) <= 5e-4     (measured 7e-5, profiles/r06_attention_p_hilo.txt)
    vision-tower flash attention, D = 80     <= 3e-3     (one bf16 P operand: hi + lo costs +25 % of that launch, same file)

Reference semantics: nn.Linear / nn.RMSNorm / mx.fast.scaled_dot_product_attention as the reference's graph types them
(mlx_vlm/models/qwen2_vl/language.py:40-154, models/base.py:305-373).  (scripts/r05_op_noise_gpu.py was the measurement; this is
the gate.)"""
import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu
BF, F64 = torch.bfloat16, torch.float64
VSLOT = [(w & 32) + 8 * (((w & 31) & 15) >> 2) + 4 * ((w & 31) >> 4) + (w & 3) for w in range(64)]
T, D, I, Hq, Hkv, hd = 64, 1536, 8960, 12, 2, 128


@pytest.fixture(scope="module")
def vops():
    from mlx_vlm_amd import ops

    return ops


def rnd(*shape, seed, s=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * s).to(BF)


def rr(a, b):
    a, b = a.detach().cpu().to(F64), b.detach().cpu().to(F64)
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def check(name, hip, orc, exact, bar):
    d_hip, d_orc = rr(hip, exact), rr(orc, exact)
    print(f"{name:60s} HIP vs exact {d_hip:.3e}   oracle vs exact {d_orc:.3e}   bar {bar:.0e}")
    assert d_hip <= bar, (name, d_hip, bar)
    return d_hip


def test_norm_and_projection_kernels_against_the_exactly_rounded_result(vops):
    x = rnd(T, D, seed=1)
    w_n = (1 + 0.1 * torch.randn(D, generator=torch.Generator().manual_seed(2))).to(BF)
    xf = x.to(F64)
    inv = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
    check("RMSNorm", vops.rmsnorm(x.cuda(), w_n.cuda()), O.rms_norm(x, w_n, 1e-6), ((xf * inv).to(BF).to(F64) * w_n.to(F64)).to(BF), 1e-4)
    W, b = rnd(2048, D, seed=3, s=0.02), rnd(2048, seed=4, s=0.1)
    check("qkv GEMM + bias (K = 1536)", vops.gemm(x.cuda(), W.cuda(), bias=b.cuda(), epilogue=vops.EPI_BIAS), O.linear(x, W, b),
          (x.to(F64) @ W.to(F64).T + b.to(F64)).to(BF), 1e-4)
    Wgu = rnd(2 * I, D, seed=5, s=0.02)
    Wg, Wu = Wgu[:I], Wgu[I:]
    Wil = torch.stack([Wg, Wu], dim=1).reshape(2 * I, D).contiguous()          # the product's interleaved (gate, up) rows
    exact = O.swiglu((x.to(F64) @ Wg.to(F64).T).to(BF), (x.to(F64) @ Wu.to(F64).T).to(BF))
    check("gate/up GEMM + SwiGLU (K = 1536)", vops.gemm(x.cuda(), Wil.cuda(), epilogue=vops.EPI_SWIGLU),
          O.swiglu(O.linear(x, Wg), O.linear(x, Wu)), exact, 1e-4)
    a, Wd, res = rnd(T, I, seed=6, s=0.3), rnd(D, I, seed=7, s=0.02), rnd(T, D, seed=8)
    exact = ((a.to(F64) @ Wd.to(F64).T).to(BF).to(F64) + res.to(F64)).to(BF)
    check("down GEMM + residual (K = 8960)", vops.gemm(a.cuda(), Wd.cuda(), res=res.cuda(), epilogue=vops.EPI_RESIDUAL),
          O.add(res, O.linear(a, Wd)), exact, 1e-4)
    # the decode step's GEMVs on one row: [RMSNorm + gate/up + SwiGLU], down + residual, the head
    x1 = rnd(1, D, seed=9)
    x1n = ((x1.to(F64) * torch.rsqrt(x1.to(F64).pow(2).mean(-1, keepdim=True) + 1e-6)).to(BF).to(F64) * w_n.to(F64)).to(BF)
    exact = O.swiglu((x1n.to(F64) @ Wg.to(F64).T).to(BF), (x1n.to(F64) @ Wu.to(F64).T).to(BF))
    xo = O.rms_norm(x1, w_n, 1e-6)
    check("decode GEMV: RMSNorm + gate/up + SwiGLU (one row)", vops.gemv(x1.cuda(), Wil.cuda(), norm_w=w_n.cuda(), epilogue=vops.EPI_SWIGLU),
          O.swiglu(O.linear(xo, Wg), O.linear(xo, Wu)), exact, 1e-4)
    a1, r1 = rnd(1, I, seed=10, s=0.3), rnd(1, D, seed=11)
    exact = ((a1.to(F64) @ Wd.to(F64).T).to(BF).to(F64) + r1.to(F64)).to(BF)
    check("decode GEMV: down + residual (K = 8960, one row)", vops.gemv(a1.cuda(), Wd.cuda(), res=r1.cuda(), epilogue=vops.EPI_RESIDUAL),
          O.add(r1, O.linear(a1, Wd)), exact, 1e-4)
    Wh = rnd(151936, D, seed=12, s=0.02)
    check("lm_head GEMV (one row, V = 151,936)", vops.gemv(x1.cuda(), Wh.cuda()), O.linear(x1, Wh), (x1.to(F64) @ Wh.to(F64).T).to(BF), 1e-4)


@pytest.mark.parametrize("Tn", [64, 384])
def test_prefill_flash_attention_against_the_exactly_rounded_result(vops, Tn):
    q, k, v = rnd(Tn, Hq, hd, seed=20), rnd(Tn, Hkv, hd, seed=21), rnd(Tn, Hkv, hd, seed=22)
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
    check(f"causal flash attention, D = 128, {Tn} tokens", out.view(Tn, Hq, hd), orc, exact, 5e-4)


def test_vision_tower_flash_attention_against_the_exactly_rounded_result(vops):
    """the ViT's attention (16 heads of 80, one 576-patch segment, not causal): ONE bf16 P operand by default"""
    Tn, H, Dv = 576, 16, 80
    q, k, v = rnd(Tn, H, Dv, seed=23), rnd(Tn, H, Dv, seed=24), rnd(Tn, H, Dv, seed=25)
    qo, ko, vo = (t.permute(1, 0, 2)[None] for t in (q, k, v))
    orc = O.sdpa(qo, ko, vo, Dv ** -0.5)[0].permute(1, 0, 2)
    exact = (torch.softmax(qo.to(F64) @ ko.to(F64).transpose(-1, -2) * Dv ** -0.5, -1) @ vo.to(F64)).to(BF)[0].permute(1, 0, 2)
    qkv = torch.cat([q.reshape(Tn, -1), k.reshape(Tn, -1), v.reshape(Tn, -1)], 1).cuda()
    cu = torch.tensor([0, Tn], dtype=torch.int32).cuda()
    out = vops.attn_prefill(qkv, qkv[:, H * Dv:], qkv[:, 2 * H * Dv:], cu, (Tn + 127) // 128, H, H, Dv, Dv ** -0.5, False)
    check("vision flash attention, D = 80, 576 patches", out.view(Tn, H, Dv), orc, exact, 3e-3)


def _paged(n, seed):
    max_pages = (n + 63) // 64 + 1
    kpool = torch.full((max_pages, Hkv, hd // 8, 64, 8), float("nan"), dtype=BF)      # unwritten slots hold NaN on purpose
    vpool = torch.full((max_pages, Hkv, hd, 64), float("nan"), dtype=BF)
    q, k, v = rnd(1, Hq * hd, seed=seed), rnd(n, Hkv, hd, seed=seed + 1), rnd(n, Hkv, hd, seed=seed + 2)
    for p in range((n + 63) // 64):
        m = min(64, n - p * 64)
        kpool[p, :, :, :m, :] = k[p * 64:p * 64 + m].permute(1, 0, 2).reshape(Hkv, m, hd // 8, 8).permute(0, 2, 1, 3)
        vpool[p][:, :, VSLOT[:m]] = v[p * 64:p * 64 + m].permute(1, 2, 0)
    rep = Hq // Hkv
    qq = q.view(Hq, 1, hd).to(F64)
    kk, vv = k.permute(1, 0, 2).to(F64).repeat_interleave(rep, 0), v.permute(1, 0, 2).to(F64).repeat_interleave(rep, 0)
    exact = (torch.softmax(qq @ kk.transpose(-1, -2) * hd ** -0.5, -1) @ vv).to(BF).reshape(1, Hq * hd)
    orc = O.sdpa(q.view(1, Hq, 1, hd), k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None], hd ** -0.5)[0, :, 0].reshape(1, Hq * hd)
    return q, kpool, vpool, max_pages, exact, orc


@pytest.mark.parametrize("n", [130, 386, 642, 2047])
def test_decode_attention_against_the_exactly_rounded_result(vops, n):
    """every form a decode step can take: one workgroup per kv head, split-K partials + combine, the page-split kernel with the
    last-arriver merge, and the one-row engine path (partials merged in the o_proj prologue - checked through an identity Wo)."""
    q, kpool, vpool, max_pages, exact, orc = _paged(n, seed=30 + n)
    qd, kd, vd = q.cuda(), kpool.cuda(), vpool.cuda()
    kv_len = torch.tensor([n], dtype=torch.int32).cuda()
    bt = torch.arange(max_pages, dtype=torch.int32)[None].cuda()
    sc = hd ** -0.5
    check(f"decode attention, one workgroup per kv head, ctx {n}", vops.attn_decode_paged(qd, kd, vd, bt, kv_len, 0, Hq, Hkv, hd, sc, 1), orc, exact, 5e-4)
    check(f"decode attention, 4 splits + combine, ctx {n}", vops.attn_decode_paged(qd, kd, vd, bt, kv_len, 0, Hq, Hkv, hd, sc, 4), orc, exact, 5e-4)
    check(f"decode attention, page-split x16, last arriver merges, ctx {n}",
          vops.attn_decode_paged_split(qd, kd, vd, None, kv_len, 0, Hq, Hkv, hd, sc, 16, max_pages=max_pages), orc, exact, 5e-4)
    po, pml = vops.attn_decode_paged_split(qd, kd, vd, None, kv_len, 0, Hq, Hkv, hd, sc, 16, max_pages=max_pages, merge=False)
    eye = torch.eye(Hq * hd, dtype=BF).cuda()
    out = vops.gemv_attn_out_bf16_(po, pml, eye, torch.zeros(1, Hq * hd, dtype=BF, device="cuda"), Hq, hd)
    check(f"decode attention, page-split partials merged in the o_proj prologue (Wo = I), ctx {n}", out, orc, exact, 5e-4)
