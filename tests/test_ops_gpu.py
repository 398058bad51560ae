"""GPU parity tests, operator level: every C-ABI entry point of libvlm_hip.so
against the oracle (oracle/ops.py) on the same seeded bf16 inputs.

Tolerances (stated per test): the HIP kernels and the oracle both accumulate in
fp32 but in different orders, so bf16 outputs agree to ~1 ulp (<= 2^-7 relative)
with a small absolute floor; attention additionally rounds P to bf16 for the
MFMA (flash-attention convention), stated below.  Integer outputs (tokens,
indices, copies) are bit-exact.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import ops as O
from tests.helpers import bf16_close

# V pool key-slot order inside a page (csrc/common.hpp vlm_vslot)
VSLOT = [(w & 32) + 8 * (((w & 31) & 15) >> 2) + 4 * ((w & 31) >> 4) + (w & 3) for w in range(64)]


def v_to_pool(v_tokens):
    """[m <= 64 tokens, Hkv, D] -> [Hkv, D, 64 slots] page image (unwritten slots NaN-free zeros)"""
    m, H, D = v_tokens.shape
    out = torch.zeros(H, D, 64, dtype=v_tokens.dtype)
    out[:, :, VSLOT[:m]] = v_tokens.permute(1, 2, 0)
    return out

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def dev():
    return torch.device("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


@pytest.fixture(scope="module")
def vops():
    from mlx_vlm_amd import ops

    return ops


def test_library_loaded_is_in_tree():
    import mlx_vlm_amd._lib as L

    assert L.lib().vlm_abi_version() == 8
    assert "mlx-vlm_amd/lib/libvlm_hip.so" in L.LIB_PATH


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(200, 512, 256), (1024, 3840, 1280), (77, 1280, 1216), (130, 1536, 8960),
                                   (5, 64, 64), (129, 136, 72), (576, 5120, 1280)])
def test_gemm_plain(vops, M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    ref = O.linear(a, w)
    out = vops.gemm(a.to(dev()), w.to(dev()))
    ok, rep = bf16_close(out, ref, ulps=2)
    assert ok, rep


def test_gemm_detects_transpose_asymmetric():
    """A = shifted identity, asymmetric W: a row/col swap in the MFMA C layout cannot pass."""
    from mlx_vlm_amd import ops

    M = N = K = 64
    a = torch.zeros(M, K)
    a[torch.arange(M), (torch.arange(M) * 7 + 3) % K] = 1.0
    w = (torch.arange(N)[:, None] * 0.5 + torch.arange(K)[None, :] * 0.01).to(BF)
    ref = O.linear(a.to(BF), w)
    out = ops.gemm(a.to(BF).to(dev()), w.to(dev()))
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("epi", ["bias", "bias_gelu_fast", "bias_gelu_erf", "bias_res", "res", "swiglu"])
def test_gemm_epilogues(vops, epi):
    M, N, K = 300, 640, 320
    a, w = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=0.08)
    b, r = rnd(N, seed=5, scale=0.5), rnd(M, N, seed=6)
    lin = O.linear(a, w, b if "bias" in epi else None)
    E = vops
    if epi == "bias":
        ref, out = lin, E.gemm(a.cuda(), w.cuda(), bias=b.cuda(), epilogue=E.EPI_BIAS)
    elif epi == "bias_gelu_fast":
        ref, out = O.gelu_fast(lin), E.gemm(a.cuda(), w.cuda(), bias=b.cuda(), epilogue=E.EPI_BIAS | E.EPI_GELU_FAST)
    elif epi == "bias_gelu_erf":
        ref, out = O.gelu_erf(lin), E.gemm(a.cuda(), w.cuda(), bias=b.cuda(), epilogue=E.EPI_BIAS | E.EPI_GELU_ERF)
    elif epi == "bias_res":
        ref, out = O.add(r, lin), E.gemm(a.cuda(), w.cuda(), bias=b.cuda(), res=r.cuda(), epilogue=E.EPI_BIAS | E.EPI_RESIDUAL)
    elif epi == "res":
        rr = r.cuda().clone()  # in place: res == out, as the engine uses it
        ref, out = O.add(r, lin), E.gemm(a.cuda(), w.cuda(), res=rr, out=rr, epilogue=E.EPI_RESIDUAL)
    else:
        g, u = O.linear(a, w[0::2]), O.linear(a, w[1::2])
        ref, out = O.swiglu(g, u), E.gemm(a.cuda(), w.cuda(), epilogue=E.EPI_SWIGLU)
    ok, rep = bf16_close(out, ref, ulps=3)
    assert ok, (epi, rep)


# ------------------------------------------------------------------ GEMV
@pytest.mark.parametrize("M", [1, 2, 4, 8])
@pytest.mark.parametrize("N,K", [(2048, 1536), (1536, 8960), (152, 256), (1000, 3584)])
def test_gemv_plain_and_bias(vops, M, N, K):
    x, w, b = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.05), rnd(N, seed=9, scale=0.3)
    out = vops.gemv(x.cuda(), w.cuda(), bias=b.cuda(), epilogue=vops.EPI_BIAS)
    ok, rep = bf16_close(out, O.linear(x, w, b), ulps=2)
    assert ok, rep


@pytest.mark.parametrize("M", [1, 4])
def test_gemv_norm_prologue_swiglu_residual(vops, M):
    K, I = 1536, 2048
    h, nw = rnd(M, K, seed=10), (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(11))).to(BF)
    wgu = rnd(2 * I, K, seed=12, scale=0.05)
    xn = O.rms_norm(h, nw, 1e-6)
    ref = O.swiglu(O.linear(xn, wgu[0::2]), O.linear(xn, wgu[1::2]))
    out = vops.gemv(h.cuda(), wgu.cuda(), norm_w=nw.cuda(), eps=1e-6, epilogue=vops.EPI_SWIGLU)
    ok, rep = bf16_close(out, ref, ulps=3)
    assert ok, rep
    wd = rnd(K, I, seed=13, scale=0.05)
    hh = h.cuda().clone()
    out2 = vops.gemv(out, wd.cuda(), res=hh, out=hh, epilogue=vops.EPI_RESIDUAL)  # in place
    ref2 = O.add(h, O.linear(out.cpu(), wd))
    ok, rep = bf16_close(out2, ref2, ulps=2)
    assert ok, rep


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("rows,dim", [(7, 160), (1024, 1280), (33, 1536), (5, 3584), (3, 8192)])
def test_layernorm_rmsnorm(vops, rows, dim):
    x = rnd(rows, dim, seed=14, scale=2.0)
    w = (1 + 0.1 * torch.randn(dim, generator=torch.Generator().manual_seed(15))).to(BF)
    b = rnd(dim, seed=16, scale=0.2)
    ok, rep = bf16_close(vops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6), O.layer_norm(x, w, b, 1e-6), ulps=2)
    assert ok, rep
    ok, rep = bf16_close(vops.rmsnorm(x.cuda(), w.cuda(), 1e-6), O.rms_norm(x, w, 1e-6), ulps=2)
    assert ok, rep
    r = rnd(rows, dim, seed=17)
    hout = torch.empty_like(x, device="cuda")
    y = vops.rmsnorm(x.cuda(), w.cuda(), 1e-6, res=r.cuda(), h_out=hout)
    hsum = O.add(x, r)
    assert torch.equal(hout.cpu(), hsum)          # residual add: exact
    ok, rep = bf16_close(y, O.rms_norm(hsum, w, 1e-6), ulps=2)
    assert ok, rep


# ------------------------------------------------------------------ rope
def test_rope2d_vision(vops):
    grid = np.array([[1, 8, 12], [1, 6, 6]])
    N, H, D = int((grid[:, 1] * grid[:, 2]).sum()), 4, 80
    qkv = rnd(N, 3 * H * D, seed=18)
    freqs = O.vision_rotary_freqs(grid, D)
    ref = qkv.clone().view(N, 3, H, D)
    ref[:, 0] = O.apply_rotary_pos_emb_vision(ref[:, 0], freqs)
    ref[:, 1] = O.apply_rotary_pos_emb_vision(ref[:, 1], freqs)
    out = vops.rope2d_vision_(qkv.cuda().clone(), torch.cos(freqs).cuda(), torch.sin(freqs).cuda(), H)
    ok, rep = bf16_close(out.view(N, 3, H, D), ref, ulps=1.01, atol_rms=1e-3)
    assert ok, rep
    assert torch.equal(out.view(N, 3, H, D)[:, 2].cpu(), qkv.view(N, 3, H, D)[:, 2])   # v untouched


def test_mrope_kvwrite_prefill_and_pages(vops):
    T, Hq, Hkv, D = 150, 4, 2, 128
    qkv = rnd(T, (Hq + 2 * Hkv) * D, seed=19)
    g = torch.Generator().manual_seed(20)
    pos = torch.randint(0, 3000, (3, T), generator=g)
    inv = O.mrope_inv_freq(D, 1e6)
    sel = O.chunked_position_selector([16, 24, 24], D // 2)
    x = qkv.view(T, Hq + 2 * Hkv, D)
    q = x[:, :Hq].permute(1, 0, 2)[None]
    k = x[:, Hq:Hq + Hkv].permute(1, 0, 2)[None]
    qr = O.mrope_apply(q, pos[:, None, :], inv, sel, "fused")[0].permute(1, 0, 2)
    kr = O.mrope_apply(k, pos[:, None, :], inv, sel, "fused")[0].permute(1, 0, 2)
    # two sequences: tokens [0,100) -> seq 0 slots 5.., tokens [100,150) -> seq 1 slots 0..
    kv_seq = torch.cat([torch.zeros(100), torch.ones(50)]).to(torch.int32)
    kv_slot = torch.cat([torch.arange(100) + 5, torch.arange(50)]).to(torch.int32)
    n_pages, max_pages = 8, 4
    bt = torch.tensor([[3, 1, 0, 0], [6, 0, 0, 0]], dtype=torch.int32)
    kpool = torch.zeros(n_pages, Hkv, D // 8, 64, 8, dtype=BF, device="cuda")
    vpool = torch.zeros(n_pages, Hkv, D, 64, dtype=BF, device="cuda")
    out = vops.mrope_kvwrite_(qkv.cuda().clone(), Hq, Hkv, D, pos[0].int().cuda(), pos[1].int().cuda(), pos[2].int().cuda(),
                              inv.cuda(), 16, 24, kv_seq.cuda(), kv_slot.cuda(), bt.cuda(), kpool, vpool)
    o = out.view(T, Hq + 2 * Hkv, D)
    ok, rep = bf16_close(o[:, :Hq], qr, ulps=1.01, atol_rms=2e-3)
    assert ok, rep
    ok, rep = bf16_close(o[:, Hq:Hq + Hkv], kr, ulps=1.01, atol_rms=2e-3)
    assert ok, rep
    assert torch.equal(o[:, Hq + Hkv:].cpu(), x[:, Hq + Hkv:])
    # page contents == what the kernel left in the qkv buffer (bit-exact copies, layout check)
    kp = kpool.cpu().permute(0, 1, 3, 2, 4).reshape(n_pages, Hkv, 64, D)
    vp = vpool.cpu()[..., VSLOT].permute(0, 1, 3, 2)      # [page, Hkv, D, slot] -> [page, Hkv, token, D]
    oc = o.cpu()
    for t in range(T):
        s, slot = int(kv_seq[t]), int(kv_slot[t])
        page, within = int(bt[s, slot // 64]), slot % 64
        assert torch.equal(kp[page, :, within], oc[t, Hq:Hq + Hkv]), t
        assert torch.equal(vp[page, :, within], oc[t, Hq + Hkv:]), t


# ------------------------------------------------------------------ attention
def _ref_attn_varlen(q, k, v, lens, scale, causal):
    """q [T,Hq,D], k/v [T,Hkv,D] -> [T,Hq,D] via oracle sdpa per segment."""
    outs, off = [], 0
    for n in lens:
        qs = q[off:off + n].permute(1, 0, 2)[None]
        ks = k[off:off + n].permute(1, 0, 2)[None]
        vs = v[off:off + n].permute(1, 0, 2)[None]
        outs.append(O.sdpa(qs, ks, vs, scale, causal=causal)[0].permute(1, 0, 2))
        off += n
    return torch.cat(outs, 0)


@pytest.mark.parametrize("D,Hq,Hkv,causal,lens", [
    (80, 4, 4, False, [100, 576, 33]),
    (80, 16, 16, False, [1024]),
    (128, 12, 2, True, [300, 64, 129]),
    (128, 4, 4, True, [1, 2, 65]),
    (64, 2, 1, False, [200]),
    (128, 2, 1, False, [130]),
])
def test_attn_prefill(vops, D, Hq, Hkv, causal, lens):
    """tolerance: P is rounded to bf16 before P.V (the oracle keeps P in fp32) -> relative 2^-8 noise
    averaged over the keys: |err| <= 2 bf16 ulps of the output + 2% of the output rms."""
    T = sum(lens)
    q, k, v = rnd(T, Hq, D, seed=21), rnd(T, Hkv, D, seed=22), rnd(T, Hkv, D, seed=23)
    scale = D ** -0.5
    ref = _ref_attn_varlen(q, k, v, lens, scale, causal)
    qkv = torch.cat([q.reshape(T, -1), k.reshape(T, -1), v.reshape(T, -1)], dim=1).cuda()
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32).cuda()
    nqb = sum((n + 127) // 128 for n in lens)
    out = vops.attn_prefill(qkv, qkv[:, Hq * D:], qkv[:, (Hq + Hkv) * D:], cu, nqb, Hq, Hkv, D, scale, causal)
    ok, rep = bf16_close(out.view(T, Hq, D), ref, ulps=2, atol_rms=2e-2)
    assert ok, rep


@pytest.mark.parametrize("D,Hq,Hkv,causal,lens,starts", [
    (128, 12, 2, True, [300, 64, 700], [257, 0, 699]),        # a prefix past two query blocks, none, all but one row
    (128, 4, 4, True, [130, 2049], [1, 2000]),
    (80, 4, 4, False, [200, 576], [100, 575]),
    (64, 2, 1, True, [300, 129], [200, 128]),
])
def test_attn_prefill_query_start_rows_before_it_are_keys_only(vops, D, Hq, Hkv, causal, lens, starts):
    """causal bit 2 / ops.attn_prefill(q_start=...) (round 6: a prompt chunk onto a non-empty cache - the cached prefix rows are
    keys only, the causal mask stays on absolute rows): the query rows equal the oracle's attention over the whole segment
    (same tolerance as test_attn_prefill) and the launch's other rows - the prefix - are left untouched."""
    T = sum(lens)
    q, k, v = rnd(T, Hq, D, seed=31), rnd(T, Hkv, D, seed=32), rnd(T, Hkv, D, seed=33)
    scale = D ** -0.5
    ref = _ref_attn_varlen(q, k, v, lens, scale, causal)
    qkv = torch.cat([q.reshape(T, -1), k.reshape(T, -1), v.reshape(T, -1)], dim=1).cuda()
    off = np.concatenate([[0], np.cumsum(lens)])
    is_q = np.zeros(T, dtype=bool)
    for i, (n, s0) in enumerate(zip(lens, starts)):
        is_q[off[i] + s0: off[i + 1]] = True
    qkv[torch.from_numpy(~is_q).cuda(), : Hq * D] = float("nan")          # the prefix rows' queries must never be read
    cu = torch.tensor(off, dtype=torch.int32).cuda()
    nqb = sum((n - s0 + 127) // 128 for n, s0 in zip(lens, starts))
    out = torch.full((T, Hq * D), 7.0, dtype=BF, device="cuda")
    vops.attn_prefill(qkv, qkv[:, Hq * D:], qkv[:, (Hq + Hkv) * D:], cu, nqb, Hq, Hkv, D, scale, causal, out=out,
                      q_start=torch.tensor(starts, dtype=torch.int32).cuda())
    o = out.cpu().view(T, Hq, D)
    ok, rep = bf16_close(o[is_q], ref[is_q], ulps=2, atol_rms=2e-2)
    assert ok, rep
    assert bool((o[~is_q].float() == 7.0).all())
    with pytest.raises(Exception):                            # not together with the uniform-placement hint
        vops.attn_prefill(qkv, qkv[:, Hq * D:], qkv[:, (Hq + Hkv) * D:], cu, nqb, Hq, Hkv, D, scale, causal, out=out,
                          q_start=torch.tensor(starts, dtype=torch.int32).cuda(), uniform_segments=True)


def test_attn_prefill_forced_rescale_spike(vops):
    """online-softmax rescale branch: one key per tile spikes the running max (guide rule 26)."""
    T, H, D = 400, 2, 128
    q, k, v = rnd(T, H, D, seed=24), rnd(T, H, D, seed=25), rnd(T, H, D, seed=26)
    for t in (70, 150, 300):   # later tiles hold much larger scores for every query
        k[t] = (q[5] * (2.0 + t / 100)).to(BF)
    ref = _ref_attn_varlen(q, k, v, [T], D ** -0.5, False)
    qkv = torch.cat([q.reshape(T, -1), k.reshape(T, -1), v.reshape(T, -1)], dim=1).cuda()
    cu = torch.tensor([0, T], dtype=torch.int32).cuda()
    out = vops.attn_prefill(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], cu, 4, H, H, D, D ** -0.5, False)
    ok, rep = bf16_close(out.view(T, H, D), ref, ulps=2, atol_rms=2e-2)
    assert ok, rep


@pytest.mark.parametrize("lens,nsplit", [([1, 130], 8), ([700, 64], 4), ([65], 1), ([2000, 1, 63, 64], 8),
                                         ([640], 1), ([513, 1100], 1), ([1, 64, 65, 2047], 1)])
def test_attn_decode_paged(vops, lens, nsplit):
    """P rounded to bf16 for the P.V MFMA (as in the prefill flash kernel): 2 ulps + 2% of the output rms."""
    B, Hq, Hkv, D = len(lens), 12, 2, 128
    scale = D ** -0.5
    g = torch.Generator().manual_seed(27)
    max_pages = max((n + 63) // 64 for n in lens)
    n_pages = sum((n + 63) // 64 for n in lens) + 3
    perm = torch.randperm(n_pages, generator=g).tolist()
    bt = torch.zeros(B, max_pages, dtype=torch.int32)
    kpool = torch.full((n_pages, Hkv, D // 8, 64, 8), float("nan"), dtype=BF)   # unwritten slots hold NaN on purpose
    vpool = torch.full((n_pages, Hkv, D, 64), float("nan"), dtype=BF)
    q = rnd(B, Hq * D, seed=28)
    refs = []
    for b, n in enumerate(lens):
        k, v = rnd(n, Hkv, D, seed=30 + b), rnd(n, Hkv, D, seed=40 + b)
        for p in range((n + 63) // 64):
            page = perm.pop()
            bt[b, p] = page
            m = min(64, n - p * 64)
            kk = k[p * 64:p * 64 + m]                       # [m, Hkv, D]
            kpool[page, :, :, :m, :] = kk.permute(1, 0, 2).reshape(Hkv, m, D // 8, 8).permute(0, 2, 1, 3)
            vpool[page][:, :, VSLOT[:m]] = v[p * 64:p * 64 + m].permute(1, 2, 0)
        qb = q[b].view(1, Hq, 1, D)
        refs.append(O.sdpa(qb, k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None], scale)[0, :, 0])
    ref = torch.stack(refs).reshape(B, Hq * D)
    kv_len = torch.tensor(lens, dtype=torch.int32)
    out = vops.attn_decode_paged(q.cuda(), kpool.cuda(), vpool.cuda(), bt.cuda(), kv_len.cuda(), 0, Hq, Hkv, D, scale, nsplit)
    ok, rep = bf16_close(out, ref, ulps=2, atol_rms=2e-2)
    assert ok, rep


@pytest.mark.parametrize("lens,nsplit,heads", [([1, 130], 8, (12, 2)), ([700, 64], 4, (12, 2)), ([65], 16, (12, 2)),
                                               ([2000, 1, 63, 64], 8, (28, 4)), ([640], 16, (12, 2)), ([513, 1100], 32, (8, 8)),
                                               ([1, 64, 65, 2047], 2, (16, 2)), ([4100], 32, (32, 8)), ([386], 16, (14, 2))])
@pytest.mark.parametrize("identity", [False, True])
def test_attn_decode_paged_split_vs_oracle(vops, lens, nsplit, heads, identity):
    """vlm_attn_decode_paged_split (one wave per page stride, the last arriver merges): every (length, split count, GQA
    group) against the oracle's SDPA - lengths of 1 token, exact page multiples, more pages than splits (a workgroup
    walks several pages), more splits than pages (workgroups that only arrive), G = 1 / 4 / 6 / 7 / 8.  The launch is
    repeated on the same ticket words: a ticket that was not re-armed would make the second launch merge early or
    never.  Same bar as the one-workgroup form: 2 ulps + 2 % of the output rms (P is rounded to bf16 for P.V)."""
    Hq, Hkv = heads
    B, D = len(lens), 128
    scale = D ** -0.5
    g = torch.Generator().manual_seed(27)
    max_pages = max((n + 63) // 64 for n in lens) + 1
    n_pages = B * max_pages if identity else sum((n + 63) // 64 for n in lens) + 3
    perm = torch.randperm(n_pages, generator=g).tolist()
    bt = torch.zeros(B, max_pages, dtype=torch.int32)
    kpool = torch.full((n_pages, Hkv, D // 8, 64, 8), float("nan"), dtype=BF)   # unwritten slots hold NaN on purpose
    vpool = torch.full((n_pages, Hkv, D, 64), float("nan"), dtype=BF)
    q = rnd(B, Hq * D, seed=28)
    refs = []
    for b, n in enumerate(lens):
        k, v = rnd(n, Hkv, D, seed=30 + b), rnd(n, Hkv, D, seed=40 + b)
        for p in range((n + 63) // 64):
            page = b * max_pages + p if identity else perm.pop()
            bt[b, p] = page
            m = min(64, n - p * 64)
            kk = k[p * 64:p * 64 + m]
            kpool[page, :, :, :m, :] = kk.permute(1, 0, 2).reshape(Hkv, m, D // 8, 8).permute(0, 2, 1, 3)
            vpool[page][:, :, VSLOT[:m]] = v[p * 64:p * 64 + m].permute(1, 2, 0)
        qb = q[b].view(1, Hq, 1, D)
        refs.append(O.sdpa(qb, k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None], scale)[0, :, 0])
    ref = torch.stack(refs).reshape(B, Hq * D)
    kv_len = torch.tensor(lens, dtype=torch.int32).cuda()
    tickets = torch.zeros(B * Hkv, dtype=torch.int32, device="cuda")
    qd, kd, vd, btd = q.cuda(), kpool.cuda(), vpool.cuda(), (None if identity else bt.cuda())
    for rep_i in range(3):
        out = vops.attn_decode_paged_split(qd, kd, vd, btd, kv_len, 0, Hq, Hkv, D, scale, nsplit, max_pages=max_pages,
                                           tickets=tickets)
        ok, rep = bf16_close(out, ref, ulps=2, atol_rms=2e-2)
        assert ok, (rep_i, rep)
        assert int(tickets.abs().sum()) == 0                      # re-armed


@pytest.mark.parametrize("n,nsplit,heads", [(386, 16, (12, 2)), (1, 16, (12, 2)), (64, 8, (12, 2)), (1024, 16, (12, 2)),
                                            (1500, 16, (16, 8)), (700, 4, (8, 1)), (130, 16, (14, 2))])
def test_attn_decode_partial_only_plus_oproj_prologue_merge(vops, n, nsplit, heads):
    """the one-row decode path of the engine: vlm_attn_decode_paged_split without merge (fp32 partials + (m, l), splits
    without a page flagged (-inf, 0); unwritten partial slots hold NaN on purpose) -> vlm_gemv_attn_out_bf16 (merge in
    the o_proj prologue + residual) against the oracle's SDPA -> Linear -> residual.  3 ulps + 2 % of the
    rms of the o_proj output (tests/test_op_noise_gpu.py holds the attention itself to 5e-4 of the exactly rounded result)."""
    Hq, Hkv = heads
    D = 128
    scale = D ** -0.5
    max_pages = (n + 63) // 64 + 2
    kpool = torch.full((max_pages, Hkv, D // 8, 64, 8), float("nan"), dtype=BF)
    vpool = torch.full((max_pages, Hkv, D, 64), float("nan"), dtype=BF)
    q = rnd(1, Hq * D, seed=61)
    k, v = rnd(n, Hkv, D, seed=62), rnd(n, Hkv, D, seed=63)
    for p in range((n + 63) // 64):
        m = min(64, n - p * 64)
        kpool[p, :, :, :m, :] = k[p * 64:p * 64 + m].permute(1, 0, 2).reshape(Hkv, m, D // 8, 8).permute(0, 2, 1, 3)
        vpool[p][:, :, VSLOT[:m]] = v[p * 64:p * 64 + m].permute(1, 2, 0)
    att = O.sdpa(q.view(1, Hq, 1, D), k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None], scale)[0, :, 0].reshape(1, Hq * D)
    N = 1536
    wo = rnd(N, Hq * D, seed=64, scale=0.03)
    h = rnd(1, N, seed=65)
    ref = (O.linear(att, wo).float() + h.float()).to(BF)
    po, pml = vops.attn_decode_paged_split(q.cuda(), kpool.cuda(), vpool.cuda(), None, torch.tensor([n], dtype=torch.int32).cuda(), 0,
                                           Hq, Hkv, D, scale, nsplit, max_pages=max_pages, merge=False)
    n_act = min(nsplit, (n + 63) // 64)
    ml = pml.cpu()
    assert bool(torch.isfinite(ml[0, :, :n_act, 0]).all()) and bool((ml[0, :, n_act:, 0] == float("-inf")).all())
    out = vops.gemv_attn_out_bf16_(po, pml, wo.cuda(), h.cuda().clone(), Hq, D)
    ok, rep = bf16_close(out, ref, ulps=3, atol_rms=2e-2)
    assert ok, rep


# ------------------------------------------------------------------ uniform 8-bit KV cache
def _q8_pools(lens, Hkv, D, seed, identity=False):
    """bf16 pools holding k / v of the given lengths on shuffled pages -> (k list, v list, bt, kpool, vpool, max_pages)"""
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    max_pages = max((n + 63) // 64 for n in lens) + 1
    n_pages = B * max_pages if identity else sum((n + 63) // 64 for n in lens) + 3
    perm = torch.randperm(n_pages, generator=g).tolist()
    bt = torch.zeros(B, max_pages, dtype=torch.int32)
    kpool = torch.zeros(n_pages, Hkv, D // 8, 64, 8, dtype=BF)
    vpool = torch.zeros(n_pages, Hkv, D, 64, dtype=BF)
    ks, vs = [], []
    for b, n in enumerate(lens):
        k, v = rnd(n, Hkv, D, seed=seed + 10 + b, scale=0.8), rnd(n, Hkv, D, seed=seed + 50 + b, scale=0.8)
        k[:, :, 5] += 3.0                                         # an outlier channel: groups with a lopsided range
        ks.append(k)
        vs.append(v)
        for p in range((n + 63) // 64):
            page = b * max_pages + p if identity else perm.pop()
            bt[b, p] = page
            m = min(64, n - p * 64)
            kpool[page, :, :, :m, :] = k[p * 64:p * 64 + m].permute(1, 0, 2).reshape(Hkv, m, D // 8, 8).permute(0, 2, 1, 3)
            vpool[page][:, :, VSLOT[:m]] = v[p * 64:p * 64 + m].permute(1, 2, 0)
    return ks, vs, bt, kpool, vpool, max_pages


def _q8_empty(kpool):
    n = kpool.numel()
    return (torch.full((n,), 77, dtype=torch.uint8, device="cuda"), torch.full((n,), 77, dtype=torch.uint8, device="cuda"),
            torch.full((n // 64,), 0x7fc07fc0, dtype=torch.int32, device="cuda"),      # unwritten (scale | bias) words: NaN | NaN
            torch.full((n // 64,), 0x7fc07fc0, dtype=torch.int32, device="cuda"))


def test_kv_quantize_tokens_bit_exact_vs_oracle(vops):
    """vlm_kv_quantize_tokens (KVCache.to_quantized over the paged pools) against the oracle's mx.quantize(bits = 8,
    group_size = 64): every u8 value and every (scale, bias) word bit for bit, tokens on shuffled pages, both groups of
    D = 128, groups whose larger-magnitude edge is the minimum and the maximum."""
    from oracle import quant as Q

    Hkv, D, lens = 2, 128, [70, 1, 129]
    ks, vs, bt, kpool, vpool, max_pages = _q8_pools(lens, Hkv, D, seed=300)
    kd, vd = kpool.cuda(), vpool.cuda()
    k8, v8, ksb, vsb = _q8_empty(kpool)
    seq = torch.cat([torch.full((n,), b, dtype=torch.int32) for b, n in enumerate(lens)])
    slot = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens])
    vops.kv_quantize_tokens(kd, vd, k8, v8, ksb, vsb, seq.cuda(), slot.cuda(), bt.cuda(), Hkv, D)
    k8v = k8.cpu().view(-1, Hkv, D // 8, 64, 8)
    v8v = v8.cpu().view(-1, Hkv, D, 64)
    ksbv, vsbv = ksb.cpu().view(-1, Hkv, 64, 2), vsb.cpu().view(-1, Hkv, 64, 2)
    for b, n in enumerate(lens):
        for name, x, pool8, sbp in (("k", ks[b], k8v, ksbv), ("v", vs[b], v8v, vsbv)):
            wq, sc, bi = Q.quantize_nd(x, 64, 8)                                   # [n, Hkv, 32 words], [n, Hkv, 2]
            ints = Q.unpack(wq.reshape(-1, wq.shape[-1]), 8).reshape(n, Hkv, D)
            want_sb = (sc.view(torch.int16).to(torch.int32) & 0xffff) | (bi.view(torch.int16).to(torch.int32) << 16)
            for t in range(n):
                page, w = int(bt[b, t // 64]), t % 64
                got = pool8[page, :, :, w, :].reshape(Hkv, D) if name == "k" else pool8[page, :, :, VSLOT[w]]
                assert torch.equal(got.to(torch.int64), ints[t]), (name, b, t)
                assert torch.equal(sbp[page, :, w, :], want_sb[t]), (name, b, t)


@pytest.mark.parametrize("lens,nsplit,heads", [([1, 130], 8, (12, 2)), ([700, 64], 4, (12, 2)), ([65], 16, (12, 2)),
                                               ([2000, 63], 8, (28, 4)), ([640], 1, (12, 2)), ([513, 1100], 2, (8, 8)),
                                               ([386], 16, (14, 2)), ([900, 900, 17], 1, (32, 32))])
@pytest.mark.parametrize("identity", [False, True])
def test_attn_decode_paged_q8_vs_oracle(vops, lens, nsplit, heads, identity):
    """vlm_attn_decode_paged_q8: the cache holds every token but the LAST one quantised (vlm_kv_quantize_tokens), the last
    one sits in the bf16 pools as the qkv epilogue leaves it - the launch quantises it (QuantizedKVCache.update_and_fetch)
    and attends over the 8-bit pools - against the oracle's quantized_scaled_dot_product_attention over its
    QuantizedKVCache (unwritten slots of the 8-bit pools hold garbage / NaN (scale, bias) words on purpose).  Launched
    twice (the second launch finds the token already quantised: same result).  2 ulps + 2 % of the rms, as the bf16
    kernels; and the 8-bit pools end up bit-identical to a to_quantized of the whole sequence."""
    from oracle import quant as Q

    Hq, Hkv = heads
    B, D = len(lens), 128
    scale = D ** -0.5
    ks, vs, bt, kpool, vpool, max_pages = _q8_pools(lens, Hkv, D, seed=400, identity=identity)
    q = rnd(B, Hq * D, seed=401)
    refs = []
    for b, n in enumerate(lens):
        c = Q.QuantizedKVCache(64, 8)
        qk, qv = c.update_and_fetch(ks[b].permute(1, 0, 2)[None], vs[b].permute(1, 0, 2)[None])
        refs.append(Q.quantized_sdpa(q[b].view(1, Hq, 1, D), qk, qv, scale)[0, :, 0])
    ref = torch.stack(refs).reshape(B, Hq * D)
    kd, vd = kpool.cuda(), vpool.cuda()
    k8, v8, ksb, vsb = _q8_empty(kpool)
    seq = torch.cat([torch.full((n - 1,), b, dtype=torch.int32) for b, n in enumerate(lens)])
    slot = torch.cat([torch.arange(n - 1, dtype=torch.int32) for n in lens])
    btd = bt.cuda()
    if seq.numel():
        vops.kv_quantize_tokens(kd, vd, k8, v8, ksb, vsb, seq.cuda(), slot.cuda(), btd, Hkv, D)
    kv_len = torch.tensor(lens, dtype=torch.int32).cuda()
    tickets = torch.zeros(B * Hkv, dtype=torch.int32, device="cuda")
    for rep_i in range(2):
        out = vops.attn_decode_paged_q8(q.cuda(), kd, vd, k8, v8, ksb, vsb, None if identity else btd, kv_len, 0, Hq, Hkv, D, scale,
                                        nsplit, quantize_new=True, max_pages=max_pages, tickets=tickets)
        ok, rep = bf16_close(out, ref, ulps=2, atol_rms=2e-2)
        assert ok, (rep_i, rep)
        assert int(tickets.abs().sum()) == 0
    k8b, v8b, ksbb, vsbb = _q8_empty(kpool)
    seq_all = torch.cat([torch.full((n,), b, dtype=torch.int32) for b, n in enumerate(lens)])
    slot_all = torch.cat([torch.arange(n, dtype=torch.int32) for n in lens])
    vops.kv_quantize_tokens(kd, vd, k8b, v8b, ksbb, vsbb, seq_all.cuda(), slot_all.cuda(), btd, Hkv, D)
    for a, b_ in ((k8, k8b), (v8, v8b), (ksb, ksbb), (vsb, vsbb)):
        assert torch.equal(a, b_)


# ------------------------------------------------------------------ gather / scatter / cast (bit-exact)
def test_embed_scatter_cast_exact(vops):
    table = rnd(1000, 256, seed=50)
    ids = torch.tensor([5, 999, 0, 17, 17], dtype=torch.int32)
    assert torch.equal(vops.embed_gather(ids.cuda(), table.cuda()).cpu(), table[ids.long()])
    dst = rnd(40, 256, seed=51).cuda()
    src = rnd(6, 256, seed=52)
    rows = torch.tensor([3, 4, 5, 20, 21, 39], dtype=torch.int32)
    ref = dst.cpu().clone()
    ref[rows.long()] = src
    assert torch.equal(vops.scatter_rows_(src.cuda(), rows.cuda(), dst).cpu(), ref)
    x = torch.randn(33, 1176, generator=torch.Generator().manual_seed(53))
    out = vops.cast_pad(x.cuda(), 1216).cpu()
    assert torch.equal(out[:, :1176], x.to(BF)) and bool((out[:, 1176:] == 0).all())


# ------------------------------------------------------------------ sampler
@pytest.mark.parametrize("B,V", [(1, 151936), (4, 1024), (2, 32000)])
def test_sample_greedy_logprobs(vops, B, V):
    logits = rnd(B, V, seed=54, scale=3.0)
    ref_lp = O.logprobs_from_logits(logits)
    tok, lp = vops.sample(logits.cuda())
    ok, rep = bf16_close(lp, ref_lp, ulps=1.01, atol_rms=0.0)   # lse in a different order: <= 1 ulp on the rounding edge
    assert ok, rep
    assert tok.cpu().tolist() == O.argmax_first(lp.cpu()).tolist()        # exact argmax of OUR logprobs, first index
    # and against the oracle's own logprobs unless they are tied/adjacent at the top
    ref_tok = O.argmax_first(ref_lp)
    for b in range(B):
        if int(tok[b]) != int(ref_tok[b]):
            assert abs(float(ref_lp[b, int(tok[b])]) - float(ref_lp[b, int(ref_tok[b])])) <= 2 ** -6 * abs(float(ref_lp[b, int(ref_tok[b])]))


def test_sample_greedy_tie_lowest_index(vops):
    logits = torch.zeros(1, 4096, dtype=BF)
    logits[0, [77, 1999, 3000]] = 5.0
    tok, _ = vops.sample(logits.cuda())
    assert int(tok[0]) == 77


def _masked_set(x):
    return set(torch.nonzero(torch.isinf(x.float()) & (x.float() < 0)).flatten().tolist())


@pytest.mark.parametrize("kind,arg", [("top_k", 50), ("top_k", 1), ("min_p", 0.05), ("top_p", 0.9), ("top_p", 0.3)])
def test_sample_filters_match_oracle(vops, kind, arg):
    """The filter masks are checked through the sampler: with the mask applied, a token outside the oracle's
    kept set must never be drawn, and over many draws every kept token with non-negligible mass appears."""
    V = 2048
    logits = rnd(1, V, seed=55, scale=2.5)
    lp = O.logprobs_from_logits(logits)
    if kind == "top_k":
        ref = O.apply_top_k(lp, arg); kw = dict(top_k=arg)
    elif kind == "min_p":
        ref = O.apply_min_p(lp, arg); kw = dict(min_p=arg)
    else:
        ref = O.apply_top_p(lp, arg); kw = dict(top_p=arg)
    kept = set(range(V)) - _masked_set(ref[0])
    # boundary elements (fp32 cumsum order / exp rounding) may go either way for top_p / min_p
    fuzzy = set()
    if kind != "top_k":
        vals = ref[0].float()
        thr_val = min(float(lp[0, i]) for i in kept)
        fuzzy = {i for i in range(V) if abs(float(lp[0, i]) - thr_val) <= 2 ** -7 * abs(thr_val) + 1e-6}
    drawn = set()
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for s in range(300):
        step.fill_(s)
        tok, _ = vops.sample(logits.cuda(), temperature=1.0, seed=123, step=step, **kw)
        drawn.add(int(tok[0]))
    assert drawn <= (kept | fuzzy), (kind, arg, sorted(drawn - kept)[:5])
    if kind == "top_k" and arg == 1:
        assert drawn == kept


def test_sample_categorical_matches_oracle_hash_rng(vops):
    """Gumbel-max with the counter hash RNG: the same (seed, step, row, index) stream as oracle/ops.py."""
    V = 4096
    logits = rnd(2, V, seed=56, scale=2.0)
    lp = O.logprobs_from_logits(logits)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    agree = 0
    for s in range(40):
        step.fill_(s)
        tok, _ = vops.sample(logits.cuda(), temperature=0.8, seed=7, step=step)
        for b in range(2):
            ref = O.categorical_gumbel(lp[b], 0.8, seed=7, step=s, row=b)
            agree += int(int(tok[b]) == ref)
    assert agree >= 78, agree   # logf/expf ulp differences may flip a near-tie


# ------------------------------------------------------------------ decode-step fusions
@pytest.mark.parametrize("M", [1, 4])
def test_gemv_qkv_rope_kvwrite_fused(vops, M):
    """[RMSNorm + qkv GEMV + bias + M-RoPE + paged KV write] vs the oracle ops chained (2 ulps: GEMV order + rope)."""
    Hq, Hkv, D, K = 12, 2, 128, 1536
    h = rnd(M, K, seed=60)
    nw = (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(61))).to(BF)
    wqkv, bqkv = rnd((Hq + 2 * Hkv) * D, K, seed=62, scale=0.05), rnd((Hq + 2 * Hkv) * D, seed=63, scale=0.3)
    pos = torch.tensor([37, 1000, 5, 2047][:M], dtype=torch.int32)
    slot = torch.tensor([70, 3, 64, 129][:M], dtype=torch.int32)
    inv = O.mrope_inv_freq(D, 1e6)
    sel = O.chunked_position_selector([16, 24, 24], D // 2)
    qkv = O.linear(O.rms_norm(h, nw, 1e-6), wqkv, bqkv).view(M, Hq + 2 * Hkv, D)
    p3 = pos.long()[None, :, None].expand(3, M, 1)
    qr = O.mrope_apply(qkv[:, :Hq][:, :, None], p3, inv, sel, "fused")[:, :, 0]
    kr = O.mrope_apply(qkv[:, Hq:Hq + Hkv][:, :, None], p3, inv, sel, "fused")[:, :, 0]
    n_pages, max_pages = 16, 4
    bt = (torch.arange(M * max_pages, dtype=torch.int32).reshape(M, max_pages) * 3 + 1) % n_pages
    kpool = torch.zeros(n_pages, Hkv, D // 8, 64, 8, dtype=BF, device="cuda")
    vpool = torch.zeros(n_pages, Hkv, D, 64, dtype=BF, device="cuda")
    out = vops.gemv_qkv_rope_kvwrite(h.cuda(), nw.cuda(), wqkv.cuda(), bqkv.cuda(), Hq, Hkv, D, pos.cuda(), slot.cuda(),
                                     inv.cuda(), bt.cuda(), kpool, vpool)
    ok, rep = bf16_close(out.view(M, Hq + 2 * Hkv, D)[:, :Hq], qr, ulps=2)
    assert ok, rep
    kp = kpool.cpu().permute(0, 1, 3, 2, 4).reshape(n_pages, Hkv, 64, D)
    vp = vpool.cpu()[..., VSLOT].permute(0, 1, 3, 2)
    for m in range(M):
        page, within = int(bt[m, int(slot[m]) // 64]), int(slot[m]) % 64
        ok, rep = bf16_close(kp[page, :, within], kr[m], ulps=2)
        assert ok, (m, rep)
        ok, rep = bf16_close(vp[page, :, within], qkv[m, Hq + Hkv:], ulps=2)
        assert ok, (m, rep)
    # nothing else was written
    assert int((kpool != 0).sum().cpu()) <= M * Hkv * D and int((vpool != 0).sum().cpu()) <= M * Hkv * D


@pytest.mark.parametrize("lens,nsplit", [([300], 2), ([700, 64, 1, 130], 3), ([5000, 999], 4)])
def test_attn_decode_partials_plus_gemv_attn_out(vops, lens, nsplit):
    """split partials -> merge fused as the o_proj GEMV prologue -> residual add, vs oracle sdpa + linear + add."""
    B, Hq, Hkv, D, N = len(lens), 12, 2, 128, 1536
    scale = D ** -0.5
    max_pages = max((n + 63) // 64 for n in lens)
    n_pages = sum((n + 63) // 64 for n in lens) + 1
    bt = torch.zeros(B, max_pages, dtype=torch.int32)
    kpool = torch.zeros(n_pages, Hkv, D // 8, 64, 8, dtype=BF)
    vpool = torch.zeros(n_pages, Hkv, D, 64, dtype=BF)
    q = rnd(B, Hq * D, seed=70)
    refs, page = [], 0
    for b, n in enumerate(lens):
        k, v = rnd(n, Hkv, D, seed=71 + b), rnd(n, Hkv, D, seed=81 + b)
        for p in range((n + 63) // 64):
            bt[b, p] = page
            m = min(64, n - p * 64)
            kpool[page, :, :, :m, :] = k[p * 64:p * 64 + m].permute(1, 0, 2).reshape(Hkv, m, D // 8, 8).permute(0, 2, 1, 3)
            vpool[page][:, :, VSLOT[:m]] = v[p * 64:p * 64 + m].permute(1, 2, 0)
            page += 1
        refs.append(O.sdpa(q[b].view(1, Hq, 1, D), k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None], scale)[0, :, 0])
    attn = torch.stack(refs).reshape(B, Hq * D)
    if B == 3:
        pytest.skip("B must be 1/2/4/8")
    wo, h = rnd(N, Hq * D, seed=90, scale=0.05), rnd(B, N, seed=91)
    ref = O.add(h, O.linear(attn, wo))
    kv_len = torch.tensor(lens, dtype=torch.int32).cuda()
    po, pml = vops.attn_decode_paged(q.cuda(), kpool.cuda(), vpool.cuda(), bt.cuda(), kv_len, 0, Hq, Hkv, D, scale, nsplit,
                                     merge=False)
    hh = h.cuda().clone()
    out = vops.gemv_attn_out_(po, pml, wo.cuda(), hh, Hq, D)
    ok, rep = bf16_close(out, ref, ulps=2, atol_rms=1e-2)
    assert ok, rep


@pytest.mark.parametrize("M,N,K", [(1024, 3840, 1280), (300, 640, 320), (77, 1280, 1216), (130, 1536, 8960)])
def test_gemm_lds_dma_staging_equals_register_staging(vops, M, N, K):
    """global_load_lds (source-swizzled DMA) and global->VGPR->LDS staging build the same LDS image: bit-identical C."""
    a, w, b = rnd(M, K, seed=100).cuda(), rnd(N, K, seed=101, scale=0.05).cuda(), rnd(N, seed=102).cuda()
    try:
        vops.gemm_set_staging(1)
        ref = vops.gemm(a, w, bias=b, epilogue=vops.EPI_BIAS | vops.EPI_GELU_FAST)
        vops.gemm_set_staging(2)      # same kernel with LDS-DMA staging (no 256 kernel, no split-K)
        out = vops.gemm(a, w, bias=b, epilogue=vops.EPI_BIAS | vops.EPI_GELU_FAST)
    finally:
        vops.gemm_set_staging(0)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("M,N,K,mode", [(386, 1536, 8960, 0), (130, 1536, 8960, 0), (200, 264, 2048, 9), (64, 640, 4096, 9)])
@pytest.mark.parametrize("epi", ["res", "bias", "bias_gelu_fast"])
def test_gemm_split_k(vops, M, N, K, mode, epi):
    """split-K (fp32 partials per K range, summed in a fixed order, epilogue in the reduce kernel): vs the oracle
    within 2 ulps (the fp32 summation order differs from the single-pass kernel), deterministic across launches;
    mode 0 = the automatic dispatch takes it for these shapes, 9 = forced x4."""
    a, w = rnd(M, K, seed=130).cuda(), rnd(N, K, seed=131, scale=0.03).cuda()
    b, r = rnd(N, seed=132).cuda(), rnd(M, N, seed=133).cuda()
    lin = O.linear(a.cpu(), w.cpu(), None if epi == "res" else b.cpu())
    if epi == "res":
        ref, kw = O.add(lin, r.cpu()), dict(res=r, epilogue=vops.EPI_RESIDUAL)
    elif epi == "bias":
        ref, kw = lin, dict(bias=b, epilogue=vops.EPI_BIAS)
    else:
        ref, kw = O.gelu_fast(lin), dict(bias=b, epilogue=vops.EPI_BIAS | vops.EPI_GELU_FAST)
    try:
        vops.gemm_set_staging(mode)
        out1 = vops.gemm(a, w, **kw)
        out2 = vops.gemm(a, w, **kw)
        vops.gemm_set_staging(8)
        single = vops.gemm(a, w, **kw)
    finally:
        vops.gemm_set_staging(0)
    assert torch.equal(out1, out2)
    ok, rep = bf16_close(out1, ref, ulps=2)
    assert ok, rep
    # vs the single-pass kernel: a 1-ulp difference of the rounded linear output can become 2 ulps after the
    # residual add / activation rounds again
    ok, rep = bf16_close(out1, single, ulps=2, atol_rms=2e-3)
    assert ok, rep


# ------------------------------------------------------------------ identity KV layout (block_table == NULL)
@pytest.mark.parametrize("lens,nsplit", [([1], 1), ([130, 64], 1), ([700, 65, 512, 3], 1), ([1500, 2048], 2)])
def test_attn_decode_identity_layout_equals_block_table_path(vops, lens, nsplit):
    """row b owns pages [b*max_pages, (b+1)*max_pages): same bits as walking an identity block table."""
    B, Hq, Hkv, D = len(lens), 12, 2, 128
    scale = D ** -0.5
    max_pages = max((n + 63) // 64 for n in lens) + 1
    kpool = (torch.randn(B * max_pages, Hkv, D // 8, 64, 8, generator=torch.Generator().manual_seed(5)) * 0.5).to(BF).cuda()
    vpool = (torch.randn(B * max_pages, Hkv, D, 64, generator=torch.Generator().manual_seed(6)) * 0.5).to(BF).cuda()
    q = rnd(B, Hq * D, seed=7).cuda()
    kv_len = torch.tensor(lens, dtype=torch.int32).cuda()
    bt = torch.arange(B * max_pages, dtype=torch.int32).reshape(B, max_pages).cuda()
    ref = vops.attn_decode_paged(q, kpool, vpool, bt, kv_len, 0, Hq, Hkv, D, scale, nsplit)
    out = vops.attn_decode_paged(q, kpool, vpool, None, kv_len, 0, Hq, Hkv, D, scale, nsplit, max_pages=max_pages)
    assert torch.equal(out, ref)


def test_gemv_qkv_kvwrite_identity_layout_equals_block_table_path(vops):
    M, Hq, Hkv, D, K = 4, 12, 2, 128, 1536
    h = rnd(M, K, seed=60).cuda()
    nw = (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(61))).to(BF).cuda()
    wqkv, bqkv = rnd((Hq + 2 * Hkv) * D, K, seed=62, scale=0.05).cuda(), rnd((Hq + 2 * Hkv) * D, seed=63, scale=0.3).cuda()
    pos = torch.tensor([37, 1000, 5, 2047], dtype=torch.int32).cuda()
    slot = torch.tensor([70, 3, 64, 129], dtype=torch.int32).cuda()
    inv = O.mrope_inv_freq(D, 1e6).cuda()
    max_pages = 4
    bt = torch.arange(M * max_pages, dtype=torch.int32).reshape(M, max_pages).cuda()
    res = []
    for table in (bt, None):
        kpool = torch.zeros(M * max_pages, Hkv, D // 8, 64, 8, dtype=BF, device="cuda")
        vpool = torch.zeros(M * max_pages, Hkv, D, 64, dtype=BF, device="cuda")
        out = vops.gemv_qkv_rope_kvwrite(h, nw, wqkv, bqkv, Hq, Hkv, D, pos, slot, inv, table, kpool, vpool,
                                         max_pages=max_pages)
        res.append((out[:, :Hq * D].clone(), kpool, vpool))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    assert int((res[1][1] != 0).sum()) > 0


@pytest.mark.parametrize("D,H,nseg,L", [(80, 16, 4, 576), (128, 8, 2, 300), (80, 4, 6, 129)])
def test_attn_prefill_uniform_segment_placement_is_bit_identical(vops, D, H, nseg, L):
    """bit 1 of `causal` only changes which workgroup computes which (segment, head, q block)."""
    T = nseg * L
    qkv = rnd(T, 3 * H * D, seed=31).cuda()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32).cuda()
    nqb = nseg * ((L + 127) // 128)
    a = vops.attn_prefill(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], cu, nqb, H, H, D, D ** -0.5, False)
    b = vops.attn_prefill(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], cu, nqb, H, H, D, D ** -0.5, False, uniform_segments=True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("causal", [False, True])
def test_attn_prefill_deferred_max_slow_ramp(vops, causal):
    """scores that creep up tile after tile by less than the deferred-max threshold (2^8), then jump: the
    reference max lags behind the true running max and P reaches ~2^8 before the rescale fires."""
    T, H, D = 640, 2, 128
    q, k, v = rnd(T, H, D, seed=34), rnd(T, H, D, seed=35), rnd(T, H, D, seed=36)
    base = q[3].float()
    for t in range(T):      # key t aligned with query 3, strength growing with t: +~1.5 (log2) per 64-key tile
        k[t] = (base * (0.02 + 0.0009 * t) / (base.norm(dim=-1, keepdim=True) / D ** 0.5)).to(BF)
    k[600] = (base * 6.0 / (base.norm(dim=-1, keepdim=True) / D ** 0.5)).to(BF)      # late spike > threshold
    ref = _ref_attn_varlen(q, k, v, [T], D ** -0.5, causal)
    qkv = torch.cat([q.reshape(T, -1), k.reshape(T, -1), v.reshape(T, -1)], dim=1).cuda()
    cu = torch.tensor([0, T], dtype=torch.int32).cuda()
    out = vops.attn_prefill(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], cu, 5, H, H, D, D ** -0.5, causal)
    ok, rep = bf16_close(out.view(T, H, D), ref, ulps=2, atol_rms=2e-2)
    assert ok, rep


# ------------------------------------------------------------------ 256x256 phased GEMM (gemm256_bf16.hip)
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 1280), (300, 520, 192), (1000, 264, 64 * 5), (9216, 1280, 1280),
                                   (777, 1536, 8960), (4100, 5120, 1280)])
@pytest.mark.parametrize("epi", ["none", "bias", "bias_gelu_fast", "bias_res", "swiglu"])
@pytest.mark.parametrize("variant", [4, 6, 7, 10])
def test_gemm256_phased_kernel_is_bit_identical_to_128_kernel(vops, M, N, K, epi, variant):
    """Same fragments, same per-element accumulation order -> the phased 256x256 schedule must reproduce the 128x128
    kernel bit for bit, including ragged M / N edges; repeated launches screen for LDS-DMA ordering races."""
    a, w = rnd(M, K, seed=110).cuda(), rnd(N, K, seed=111, scale=0.05).cuda()
    b, r = rnd(N, seed=112).cuda(), rnd(M, N, seed=113).cuda()
    kw = {"none": dict(), "bias": dict(bias=b, epilogue=vops.EPI_BIAS),
          "bias_gelu_fast": dict(bias=b, epilogue=vops.EPI_BIAS | vops.EPI_GELU_FAST),
          "bias_res": dict(bias=b, res=r, epilogue=vops.EPI_BIAS | vops.EPI_RESIDUAL),
          "swiglu": dict(epilogue=vops.EPI_SWIGLU)}[epi]
    if epi == "swiglu" and N % 16:
        pytest.skip("SwiGLU pairs need N % 16 == 0")
    try:
        vops.gemm_set_staging(2)
        ref = vops.gemm(a, w, **kw)
        vops.gemm_set_staging(variant)     # 4: two phases of 32 MFMAs; 6 / 7: four phases of 16, 256x192 / 256x256 tiles; 10: persistent tile loop
        for it in range(6):
            out = vops.gemm(a, w, **kw)
            assert torch.equal(out, ref), f"iteration {it}: {int((out != ref).sum())} elements differ"
    finally:
        vops.gemm_set_staging(0)


@pytest.mark.parametrize("variant", [4, 6, 7, 10])
def test_gemm256_under_memory_pressure_race_screen(vops, variant):
    """a concurrent copy stream perturbs DMA landing order; results must not change"""
    M, N, K = (2048, 2048, 2048) if variant != 10 else (6144, 4096, 1024)     # 10: > 256 tiles -> persistent loop
    a, w = rnd(M, K, seed=120).cuda(), rnd(N, K, seed=121, scale=0.05).cuda()
    big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(big)
    try:
        vops.gemm_set_staging(2)
        ref = vops.gemm(a, w)
        vops.gemm_set_staging(variant)
        side = torch.cuda.Stream()
        for it in range(8):
            with torch.cuda.stream(side):
                dst.copy_(big, non_blocking=True)
            out = vops.gemm(a, w)
            torch.cuda.synchronize()
            assert torch.equal(out, ref), f"iteration {it}"
    finally:
        vops.gemm_set_staging(0)


@pytest.mark.parametrize("B,V", [(1, 151936), (3, 4099), (8, 32000)])
def test_sample_greedy_advance_equals_sample_plus_advance_plus_gather(vops, B, V):
    """The fused greedy tail (2 launches) against the unfused sequence it replaces: vlm_sample (temperature 0) +
    vlm_decode_advance + the next step's vlm_embed_gather - every output bit-identical, repeatedly on the same
    workspace (the arrival ticket re-arms itself)."""
    from mlx_vlm_amd import _lib
    from mlx_vlm_amd._lib import check
    import ctypes as C

    dev = "cuda"
    D = 64
    torch.manual_seed(B + V)
    embed = torch.randn(V, D, device=dev).to(BF)
    ws = vops.sample_workspace(B, dev)
    ring_len = 4
    state = lambda: dict(ctx=torch.arange(10, 10 + B, dtype=torch.int32, device=dev),   # noqa: E731
                         pos=torch.arange(20, 20 + B, dtype=torch.int32, device=dev),
                         step=torch.zeros(1, dtype=torch.int32, device=dev),
                         ring=torch.full((ring_len, B), -1, dtype=torch.int32, device=dev))
    a, b = state(), state()
    tok_a = torch.zeros(B, dtype=torch.int32, device=dev)
    h_a = torch.zeros(B, D, dtype=BF, device=dev)
    for it in range(6):
        logits = (torch.randn(B, V, device=dev) * 3).to(BF)
        if it == 3:                                    # exact ties across blocks: the lowest index must win
            logits[:, V // 3] = 50.0
            logits[:, V - 5] = 50.0
        lp_a = vops.sample_greedy_advance(logits, tok_a, a["ctx"], a["pos"], a["step"], embed, h_a, out_ring=a["ring"], ws=ws)
        tok_b, lp_b = vops.sample(logits, step=b["step"])
        check(_lib.lib().vlm_decode_advance(C.c_void_p(b["ctx"].data_ptr()), C.c_void_p(b["pos"].data_ptr()),
                                            C.c_void_p(tok_b.data_ptr()), C.c_void_p(b["ring"].data_ptr()), ring_len,
                                            C.c_void_p(b["step"].data_ptr()), B,
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "advance")
        h_b = vops.embed_gather(tok_b, embed)
        assert torch.equal(tok_a, tok_b), it
        assert torch.equal(lp_a, lp_b)
        assert torch.equal(h_a, h_b)
        for k in ("ctx", "pos", "step", "ring"):
            assert torch.equal(a[k], b[k]), (it, k)
    assert int(a["step"][0]) == 6


def test_sample_greedy_advance_nan_row_gives_token_zero_not_a_fault(vops):
    """A logits row without a finite candidate (all NaN) must not turn the argmax sentinel into an embedding address (BENCH_r04:
    a GPU memory fault kills the process): token 0 as the unfused kernels give, the event counted, the other rows and the
    next call unaffected."""
    dev = "cuda"
    B, V, D = 3, 4099, 64
    torch.manual_seed(7)
    embed = torch.randn(V, D, device=dev).to(BF)
    ws = vops.sample_workspace(B, dev)
    ctx = torch.arange(10, 10 + B, dtype=torch.int32, device=dev)
    pos = torch.arange(20, 20 + B, dtype=torch.int32, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    tok = torch.full((B,), -7, dtype=torch.int32, device=dev)
    h = torch.zeros(B, D, dtype=BF, device=dev)
    logits = (torch.randn(B, V, device=dev) * 3).to(BF)
    good = logits.clone()
    logits[1] = float("nan")
    vops.sample_greedy_advance(logits, tok, ctx, pos, step, embed, h, ws=ws)
    torch.cuda.synchronize()
    ref, _ = vops.sample(good, step=torch.zeros(1, dtype=torch.int32, device=dev))
    assert int(tok[1]) == 0 and int(tok[0]) == int(ref[0]) and int(tok[2]) == int(ref[2])
    assert torch.equal(h[1], embed[0]) and torch.equal(h[0], embed[int(ref[0])])
    assert vops.bad_argmax_rows(ws) == 1
    vops.sample_greedy_advance(good, tok, ctx, pos, step, embed, h, ws=ws)          # the ticket re-armed itself
    assert torch.equal(tok, ref) and vops.bad_argmax_rows(ws) == 1 and int(step[0]) == 2


def test_logit_penalties_kernel_bit_exact_vs_oracle_and_reference_golden(vops):
    """vlm_apply_logit_penalties (bias -> repetition -> presence -> frequency over the device token history) against the
    oracle on the reference-generated cases (bit-exact bf16), then the push path: tokens appended one by one, ring
    wrap-around included, equal to the oracle on the growing history."""
    import os

    from mlx_vlm_amd import _lib
    from mlx_vlm_amd.sample_utils import HIST_CAP

    P = np.load(os.path.join(os.path.dirname(__file__), "golden", "penalties_ref.npz"))
    dev = "cuda"
    for ci in range(int(P["n_cases"])):
        g = lambda k: P[f"case{ci}.{k}"]                                      # noqa: E731
        f = lambda k: 0.0 if np.isnan(g(k)) else float(g(k))                   # noqa: E731
        toks = g("tokens")
        x = torch.from_numpy(g("logits_bf16_as_f32")).to(BF).to(dev)
        B = x.shape[0]
        hist = torch.zeros(B, HIST_CAP, dtype=torch.int32, device=dev)
        hist[:, :len(toks)] = torch.from_numpy(toks.astype(np.int32)).to(dev)
        hlen = torch.full((B,), len(toks), dtype=torch.int32, device=dev)
        bi = torch.from_numpy(g("bias_idx").astype(np.int32)).to(dev)
        bv = torch.from_numpy(g("bias_val").astype(np.float32)).to(dev)
        pa = _lib.PenaltyArgs(hist.data_ptr(), hlen.data_ptr(), HIST_CAP, f("rep"), int(g("rep_ctx")), f("pres"),
                              int(g("pres_ctx")), f("freq"), int(g("freq_ctx")), bi.data_ptr() if len(bi) else None,
                              bv.data_ptr() if len(bv) else None, len(bi))
        vops.apply_logit_penalties(x, pa)
        assert torch.equal(x.cpu(), torch.from_numpy(g("out_bf16_as_f32")).to(BF)), ci
    # push path with wrap-around: cap 256, 300 pushes
    rng = np.random.default_rng(3)
    V = 500
    hist = torch.zeros(1, HIST_CAP, dtype=torch.int32, device=dev)
    hlen = torch.zeros(1, dtype=torch.int32, device=dev)
    pa = _lib.PenaltyArgs(hist.data_ptr(), hlen.data_ptr(), HIST_CAP, 1.25, 40, 0.5, 256, 0.125, 7, None, None, 0)
    fed = []
    for step in range(300):
        t = int(rng.integers(0, 30))
        fed.append(t)
        logits = (torch.randn(1, V) * 2).to(BF)
        ref = O.apply_logits_processors(logits, fed, None, 1.25, 40, 0.5, 256, 0.125, 7)
        x = logits.to(dev)
        vops.apply_logit_penalties(x, pa, push_tok=torch.tensor([t], dtype=torch.int32, device=dev))
        if step % 23 == 0 or step > 290:
            assert torch.equal(x.cpu(), ref), step
    assert int(hlen[0]) == 300


@pytest.mark.parametrize("M,H,hd", [(9216, 16, 80), (1024, 16, 80), (200, 2, 80), (77, 3, 64)])
def test_gemm_rope2d_epilogue_equals_gemm_then_rope_pass(vops, M, H, hd):
    """vlm_gemm_bf16_rope2d (2-D rope of the vision attention in the qkv GEMM epilogue, q / k rows interleaved per head)
    against the two-pass form it replaces - vlm_gemm_bf16 + bias, then vlm_rope2d_vision - on the original row order:
    same contraction, same bias rounding, same rotation formula; the outputs are compared after undoing the
    interleave.  Shapes cover the 256-tile kernel (16 x 336^2 images), the 128-tile kernel and ragged edges."""
    import ctypes as C

    from mlx_vlm_amd import _lib

    dev = "cuda"
    E = H * hd
    K = 1280 if hd == 80 else 192
    torch.manual_seed(M + H)
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    w = (torch.randn(3 * E, K, device=dev) * 0.05).to(BF)
    b = (torch.randn(3 * E, device=dev) * 0.1).to(BF)
    freqs = torch.rand(M, hd // 2, device=dev) * 6.0
    cs = torch.stack([torch.cos(freqs), torch.sin(freqs)]).contiguous()
    # reference form
    ref = vops.gemm(a, w, bias=b, epilogue=vops.EPI_BIAS)
    check = _lib.check
    check(_lib.lib().vlm_rope2d_vision(C.c_void_p(ref.data_ptr()), C.c_void_p(cs[0].data_ptr()), C.c_void_p(cs[1].data_ptr()),
                                       M, H, hd, 3 * E, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rope2d")
    # fused form on interleaved rows
    half = hd // 2
    inter = torch.stack([torch.arange(half), torch.arange(half) + half], dim=1).reshape(-1)
    perm = torch.cat([(torch.arange(2 * H)[:, None] * hd + inter[None, :]).reshape(-1), torch.arange(2 * E, 3 * E)]).to(dev)
    out = vops.gemm_rope2d(a, w[perm].contiguous(), b[perm].contiguous(), cs, hd, 2 * E)
    got = torch.empty_like(out)
    got[:, perm] = out                                   # undo the interleave
    assert torch.equal(got[:, 2 * E:], ref[:, 2 * E:])   # v: bias only, untouched
    exact = float((got == ref).float().mean())
    ok, rep = bf16_close(got, ref, ulps=1, atol_rms=0.0)
    assert ok and exact > 0.999, (exact, rep)            # fp32 contraction of a*c - b*s may differ in the last place


# ------------------------------------------------------------------ MLX affine 4-bit weights (csrc/gemv_w4.hip)
def _q4(N, K, seed, scale=0.05):
    """-> (oracle QW, device QuantW) of one seeded matrix quantized the way mx.quantize does (oracle/quant.py)"""
    from mlx_vlm_amd.models import quantized as Qz
    from oracle import quant as Q

    w = rnd(N, K, seed=seed, scale=scale)
    wq, s, b = Q.quantize_affine(w)
    dq = Qz.take({"p.weight": wq.view(torch.uint32), "p.scales": s, "p.biases": b}, "p")
    return Q.QW(wq, s, b), dq.to("cuda")


@pytest.mark.parametrize("N,K", [(96, 256), (33, 1536), (7, 8960)])
def test_dequant_w4_bit_exact_and_row_gather(vops, N, K):
    """vlm_dequant_w4 == mx.dequantize restated (fp32 scale * q + bias, one rounding): BIT-EXACT, whole matrix and a
    gather with repeats (nn.QuantizedEmbedding lookup)."""
    from oracle import quant as Q

    ow, dw = _q4(N, K, seed=200 + N)
    ref = Q.dequantize(ow.wq, ow.scales, ow.biases)
    assert torch.equal(vops.dequant_w4(dw.wq, dw.sb).cpu(), ref)
    idx = torch.tensor([N - 1, 0, 5 % N, N - 1, 2 % N], dtype=torch.int32)
    assert torch.equal(vops.dequant_w4(dw.wq, dw.sb, rows=idx.cuda()).cpu(), ref[idx.long()])
    assert torch.equal(ow.rows(idx), ref[idx.long()])


@pytest.mark.parametrize("M,N,K", [(200, 512, 256), (885, 3072, 3072), (130, 1536, 8960), (77, 1280, 1216 + 64), (5, 64, 64),
                                   (1400, 8192, 1536), (386, 3072, 8192)])
def test_gemm_w4_fused_bit_identical_to_dequant_then_gemm(vops, M, N, K):
    """vlm_gemm_w4 (BASELINE configs[4]: the dequant-fused prefill GEMM) against dequantise-then-GEMM (vlm_dequant_w4 +
    vlm_gemm_bf16): BIT-IDENTICAL - the fused kernel builds the same LDS image of every W tile - for every tile shape
    (128x128, 64x128, 64x64), the split-K form (few tiles x long K: the down projection at prompt length), ragged M, and
    the epilogues of the decoder (bias, residual, SwiGLU on interleaved rows); and against the oracle's
    nn.QuantizedLinear (fp32 dequantised weights)."""
    ow, dw = _q4(N, K, seed=500 + M)
    a, b, r = rnd(M, K, seed=501), rnd(N, seed=502, scale=0.3), rnd(M, N, seed=503)
    wd = vops.dequant_w4(dw.wq, dw.sb)
    ad = a.cuda()
    assert torch.equal(vops.gemm_w4(ad, dw.wq, dw.sb), vops.gemm(ad, wd))
    assert torch.equal(vops.gemm_w4(ad, dw.wq, dw.sb, bias=b.cuda(), epilogue=vops.EPI_BIAS),
                       vops.gemm(ad, wd, bias=b.cuda(), epilogue=vops.EPI_BIAS))
    assert torch.equal(vops.gemm_w4(ad, dw.wq, dw.sb, res=r.cuda(), epilogue=vops.EPI_RESIDUAL),
                       vops.gemm(ad, wd, res=r.cuda(), epilogue=vops.EPI_RESIDUAL))
    if N % 16 == 0:
        assert torch.equal(vops.gemm_w4(ad, dw.wq, dw.sb, epilogue=vops.EPI_SWIGLU), vops.gemm(ad, wd, epilogue=vops.EPI_SWIGLU))
    # against the oracle: the prefill forms multiply by the bf16-ROUNDED dequantised weights (mx.dequantize's output dtype;
    # the reference's quantized_matmul keeps them in fp32) - each weight off by up to 2^-9 relative, the sum over K of those
    # independent errors is ~0.2 % of the output rms, a few of 10^5..10^7 outputs reach 5 sigma: 2 ulps + 2 % of the rms
    ok, rep = bf16_close(vops.gemm_w4(ad, dw.wq, dw.sb), ow.linear(a), ulps=2, atol_rms=2e-2)
    assert ok, rep


@pytest.mark.parametrize("M,N,K", [(1, 512, 256), (2, 1536, 1536), (8, 8192, 1536), (4, 1536, 8960), (1, 302, 4096),
                                   (1, 1024, 18944)])
def test_gemv_w4_plain_bias_residual_vs_oracle(vops, M, N, K):
    """vlm_gemv_w4 vs nn.QuantizedLinear restated (oracle/quant.py::quantized_linear): every (R, KC) instantiation
    (K <= 2048 / 4096 / 10240 / 20480 chunks per lane; 4 rows per wave at N >= 8192), ragged N.  2 ulps + 2e-3 rms:
    both sides accumulate the exact fp32 affine form, in different orders."""
    ow, dw = _q4(N, K, seed=210 + M)
    x, b, r = rnd(M, K, seed=211), rnd(N, seed=212, scale=0.3), rnd(M, N, seed=213)
    ok, rep = bf16_close(vops.gemv_w4(x.cuda(), dw.wq, dw.sb), ow.linear(x), ulps=2)
    assert ok, rep
    ok, rep = bf16_close(vops.gemv_w4(x.cuda(), dw.wq, dw.sb, bias=b.cuda(), epilogue=vops.EPI_BIAS), ow.linear(x, b), ulps=2)
    assert ok, rep
    rr = r.cuda().clone()
    out = vops.gemv_w4(x.cuda(), dw.wq, dw.sb, res=rr, out=rr, epilogue=vops.EPI_RESIDUAL)          # in place
    ok, rep = bf16_close(out, O.add(r, ow.linear(x)), ulps=2)
    assert ok, rep


@pytest.mark.parametrize("M", [1, 4])
def test_gemv_w4_norm_prologue_swiglu_and_head(vops, M):
    """the decode MLP half over 4-bit weights: [RMSNorm + gate/up + SwiGLU] on interleaved rows, then the norm + lm_head
    form (N >= 8192: 4 rows per wave)."""
    from mlx_vlm_amd.models import quantized as Qz

    K, I = 1536, 2048
    h, nw = rnd(M, K, seed=220), (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(221))).to(BF)
    og, dg = _q4(I, K, seed=222)
    ou, du = _q4(I, K, seed=223)
    dgu = Qz.interleave_rows(dg, du)
    xn = O.rms_norm(h, nw, 1e-6)
    ref = O.swiglu(og.linear(xn), ou.linear(xn))
    out = vops.gemv_w4(h.cuda(), dgu.wq, dgu.sb, norm_w=nw.cuda(), eps=1e-6, epilogue=vops.EPI_SWIGLU)
    ok, rep = bf16_close(out, ref, ulps=3)
    assert ok, rep
    oh, dh = _q4(9000, K, seed=224)
    ok, rep = bf16_close(vops.gemv_w4(h.cuda(), dh.wq, dh.sb, norm_w=nw.cuda(), eps=1e-6), oh.linear(xn), ulps=2)
    assert ok, rep


@pytest.mark.parametrize("M", [1, 2])
def test_gemv_w4_qkv_rope_kvwrite_fused(vops, M):
    """[RMSNorm + 4-bit qkv GEMV + bias + M-RoPE + paged KV write]: the bf16 test above with quantized q / k / v rows
    concatenated the way the loader packs them; the bias add is a second typed op after the matmul's rounding."""
    from mlx_vlm_amd.models import quantized as Qz

    Hq, Hkv, D, K = 12, 2, 128, 1536
    h = rnd(M, K, seed=60)
    nw = (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(61))).to(BF)
    parts = [_q4(Hq * D, K, seed=230), _q4(Hkv * D, K, seed=231), _q4(Hkv * D, K, seed=232)]
    dq = Qz.cat_rows([p[1] for p in parts])
    bqkv = rnd((Hq + 2 * Hkv) * D, seed=63, scale=0.3)
    pos = torch.tensor([37, 1000], dtype=torch.int32)[:M]
    slot = torch.tensor([70, 3], dtype=torch.int32)[:M]
    inv = O.mrope_inv_freq(D, 1e6)
    sel = O.chunked_position_selector([16, 24, 24], D // 2)
    xn = O.rms_norm(h, nw, 1e-6)
    bs = [bqkv[:Hq * D], bqkv[Hq * D:(Hq + Hkv) * D], bqkv[(Hq + Hkv) * D:]]
    qkv = torch.cat([p[0].linear(xn, b) for p, b in zip(parts, bs)], -1).view(M, Hq + 2 * Hkv, D)
    p3 = pos.long()[None, :, None].expand(3, M, 1)
    qr = O.mrope_apply(qkv[:, :Hq][:, :, None], p3, inv, sel, "fused")[:, :, 0]
    kr = O.mrope_apply(qkv[:, Hq:Hq + Hkv][:, :, None], p3, inv, sel, "fused")[:, :, 0]
    n_pages, max_pages = 16, 4
    bt = (torch.arange(M * max_pages, dtype=torch.int32).reshape(M, max_pages) * 3 + 1) % n_pages
    kpool = torch.zeros(n_pages, Hkv, D // 8, 64, 8, dtype=BF, device="cuda")
    vpool = torch.zeros(n_pages, Hkv, D, 64, dtype=BF, device="cuda")
    out = vops.gemv_w4_qkv_rope_kvwrite(h.cuda(), nw.cuda(), dq.wq, dq.sb, bqkv.cuda(), Hq, Hkv, D, pos.cuda(), slot.cuda(),
                                        inv.cuda(), bt.cuda(), kpool, vpool)
    ok, rep = bf16_close(out.view(M, Hq + 2 * Hkv, D)[:, :Hq], qr, ulps=2)
    assert ok, rep
    kp = kpool.cpu().permute(0, 1, 3, 2, 4).reshape(n_pages, Hkv, 64, D)
    vp = vpool.cpu()[..., VSLOT].permute(0, 1, 3, 2)
    for m in range(M):
        page, within = int(bt[m, int(slot[m]) // 64]), int(slot[m]) % 64
        ok, rep = bf16_close(kp[page, :, within], kr[m], ulps=2)
        assert ok, (m, rep)
        ok, rep = bf16_close(vp[page, :, within], qkv[m, Hq + Hkv:], ulps=2)
        assert ok, (m, rep)
    assert int((kpool != 0).sum().cpu()) <= M * Hkv * D and int((vpool != 0).sum().cpu()) <= M * Hkv * D


# ------------------------------------------------------------------ batched decode rows on the matrix cores (csrc/gemv_mfma.hip)
@pytest.mark.parametrize("M", [5, 8, 13, 16])
@pytest.mark.parametrize("H,I", [(1536, 8960), (1024, 4096), (2048, 5632), (896, 4864)])
def test_gemv_tiled_mlp_pair_equals_the_plain_pair(vops, M, H, I):
    """The MLP of a batched decode step with the activations handed over in the MFMA tile's layout (VLM_EPI_Y_TILED on the
    [RMSNorm + gate/up + SwiGLU] launch, VLM_EPI_X_TILED on the down projection = its row-slice form, csrc/gemv_mfma_rows.hip:
    one workgroup per CU owns N / 256 output rows and their whole K) against the plain pair and the oracle: the SwiGLU output is
    the SAME values at other addresses (bit-exact, rows M..15 never written), the down projection sums in another fp32 order
    (2 ulps vs the oracle, as every form), repeats are bit-identical, ragged slices (N = 896: 4 rows per CU, the last ones short)."""
    x, nw = rnd(M, H, seed=70), (1 + 0.1 * torch.randn(H, generator=torch.Generator().manual_seed(71))).to(BF)
    wg, wu, wd = rnd(I, H, seed=72, scale=0.05), rnd(I, H, seed=73, scale=0.05), rnd(H, I, seed=74, scale=0.03)
    wil = torch.stack([wg, wu], dim=1).reshape(2 * I, H).contiguous().cuda()
    xd, nwd, wdd = x.cuda(), nw.cuda(), wd.cuda()
    act = vops.gemv_ws(xd, wil, norm_w=nwd, epilogue=vops.EPI_SWIGLU)
    act_t = vops.gemv_ws(xd, wil, norm_w=nwd, epilogue=vops.EPI_SWIGLU | vops.EPI_Y_TILED)
    assert act_t.shape == (I // 8, 16, 8)
    assert torch.equal(vops.untile_rows(act_t, M), act)
    if M < 16:
        assert bool(torch.isnan(act_t[:, M:].float()).all())          # rows past M are not written
    r = rnd(M, H, seed=75)
    want = vops.gemv_ws(act, wdd, res=r.cuda(), epilogue=vops.EPI_RESIDUAL)
    got = vops.gemv_ws(act_t, wdd, res=r.cuda(), epilogue=vops.EPI_RESIDUAL | vops.EPI_X_TILED, M=M)
    ref = O.add(r, O.linear(act.cpu(), wd))
    ok, rep = bf16_close(got, ref, ulps=2)
    assert ok, rep
    ok, rep = bf16_close(got, want.cpu(), ulps=2)
    assert ok, rep
    for _ in range(3):
        assert torch.equal(vops.gemv_ws(act_t, wdd, res=r.cuda(), epilogue=vops.EPI_RESIDUAL | vops.EPI_X_TILED, M=M), got)
    # bias + residual, no epilogue; a tiled x built by hand
    b = rnd(H, seed=76, scale=0.3)
    a2 = rnd(M, I, seed=77, scale=0.3)
    ok, rep = bf16_close(vops.gemv_ws(vops.tile_rows(a2.cuda()), wdd, bias=b.cuda(), epilogue=vops.EPI_BIAS | vops.EPI_X_TILED, M=M),
                         O.linear(a2, wd, b), ulps=2)
    assert ok, rep
    ok, rep = bf16_close(vops.gemv_ws(vops.tile_rows(a2.cuda()), wdd, epilogue=vops.EPI_X_TILED, M=M), O.linear(a2, wd), ulps=2)
    assert ok, rep


def test_gemv_tiled_flags_refuse_what_the_tiled_kernels_do_not_take(vops):
    """a flagged call whose shape the matrix-core forms do not take is refused (VLM_ERR_SHAPE) and enqueues nothing"""
    x, w = rnd(8, 1536, seed=78).cuda(), rnd(1536, 1536, seed=79, scale=0.05).cuda()
    with pytest.raises(RuntimeError):
        vops.gemv_ws(vops.tile_rows(x), w, epilogue=vops.EPI_X_TILED, M=8)            # K < 4096: not the row-slice form's
    x2 = rnd(2, 1536, seed=80).cuda()
    with pytest.raises(RuntimeError):
        vops.gemv_ws(x2, w, epilogue=vops.EPI_Y_TILED)                                # 2 rows: the v_dot2c kernels, which do not tile


@pytest.mark.parametrize("M", [5, 8, 13, 16])
@pytest.mark.parametrize("N,K", [(2048, 1536), (1536, 8960), (1000, 3584), (152, 256), (3584, 18944)])
def test_gemv_mfma_rows_plain_bias_residual(vops, M, N, K):
    """5..16 batch rows as the N dimension of the MFMA (vlm_gemv_bf16_ws): every K-split form (one segment; 6 segments of
    the 8960-wide down projection through the workspace + tickets; 13 of the 7B one), ragged N, bias, residual in place,
    bias + residual - vs the oracle (2 ulps: fp32 accumulation in another order), and bit-identical when repeated (the
    partial tiles are summed in a fixed order)."""
    x, w, b = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.05), rnd(N, seed=9, scale=0.3)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    ref = O.linear(x, w)
    out = vops.gemv_ws(xd, wd)
    ok, rep = bf16_close(out, ref, ulps=2)
    assert ok, rep
    for _ in range(3):
        assert torch.equal(vops.gemv_ws(xd, wd), out)
    ok, rep = bf16_close(vops.gemv_ws(xd, wd, bias=bd, epilogue=vops.EPI_BIAS), O.linear(x, w, b), ulps=2)
    assert ok, rep
    r = rnd(M, N, seed=10)
    rr = r.cuda().clone()
    out2 = vops.gemv_ws(xd, wd, res=rr, out=rr, epilogue=vops.EPI_RESIDUAL)
    ok, rep = bf16_close(out2, O.add(r, ref), ulps=2)
    assert ok, rep


@pytest.mark.parametrize("M", [6, 8, 13, 16])
@pytest.mark.parametrize("K,I", [(1536, 2048), (3584, 1024), (896, 256), (3072, 512), (4096, 512)])
def test_gemv_mfma_rows_norm_prologue_swiglu_and_head(vops, M, K, I):
    """(every activation prologue of the kernel: everything in registers at <= 8 rows x K <= 1536, the row-trip forms with
    3 / 8 chunks per lane and 1 / 2 trips above that, rows that end inside a lane's chunk range - K = 896, 3072)"""
    h, nw = rnd(M, K, seed=10), (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(11))).to(BF)
    wgu = rnd(2 * I, K, seed=12, scale=0.05)
    xn = O.rms_norm(h, nw, 1e-6)
    ref = O.swiglu(O.linear(xn, wgu[0::2]), O.linear(xn, wgu[1::2]))
    out = vops.gemv_ws(h.cuda(), wgu.cuda(), norm_w=nw.cuda(), eps=1e-6, epilogue=vops.EPI_SWIGLU)
    ok, rep = bf16_close(out, ref, ulps=3)
    assert ok, rep
    wh = rnd(9000, K, seed=13, scale=0.05)
    ok, rep = bf16_close(vops.gemv_ws(h.cuda(), wh.cuda(), norm_w=nw.cuda(), eps=1e-6), O.linear(xn, wh), ulps=2)
    assert ok, rep


@pytest.mark.parametrize("M", [9, 16])
@pytest.mark.parametrize("paged", [True, False])
def test_gemv_mfma_rows_qkv_rope_kvwrite(vops, M, paged):
    """[RMSNorm + qkv + bias + M-RoPE + paged KV write] for 9..16 rows: a tile's 16 rows are (d0..d0+7, d0+64..d0+71) of one
    head so both halves of a rotation pair meet in one workgroup; block-table and identity layouts."""
    Hq, Hkv, D, K = 12, 2, 128, 1536
    h = rnd(M, K, seed=60)
    nw = (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(61))).to(BF)
    wqkv, bqkv = rnd((Hq + 2 * Hkv) * D, K, seed=62, scale=0.05), rnd((Hq + 2 * Hkv) * D, seed=63, scale=0.3)
    g = torch.Generator().manual_seed(64)
    pos = torch.randint(0, 3000, (M,), generator=g, dtype=torch.int32)
    max_pages = 4
    slot = torch.randint(0, 64 * max_pages, (M,), generator=g, dtype=torch.int32)
    inv = O.mrope_inv_freq(D, 1e6)
    sel = O.chunked_position_selector([16, 24, 24], D // 2)
    qkv = O.linear(O.rms_norm(h, nw, 1e-6), wqkv, bqkv).view(M, Hq + 2 * Hkv, D)
    p3 = pos.long()[None, :, None].expand(3, M, 1)
    qr = O.mrope_apply(qkv[:, :Hq][:, :, None], p3, inv, sel, "fused")[:, :, 0]
    kr = O.mrope_apply(qkv[:, Hq:Hq + Hkv][:, :, None], p3, inv, sel, "fused")[:, :, 0]
    n_pages = M * max_pages
    bt = torch.randperm(n_pages, generator=g).to(torch.int32).reshape(M, max_pages) if paged else \
        torch.arange(n_pages, dtype=torch.int32).reshape(M, max_pages)
    kpool = torch.zeros(n_pages, Hkv, D // 8, 64, 8, dtype=BF, device="cuda")
    vpool = torch.zeros(n_pages, Hkv, D, 64, dtype=BF, device="cuda")
    out = vops.gemv_qkv_rope_kvwrite_ws(h.cuda(), nw.cuda(), wqkv.cuda(), bqkv.cuda(), Hq, Hkv, D, pos.cuda(), slot.cuda(),
                                        inv.cuda(), bt.cuda() if paged else None, kpool, vpool, max_pages=max_pages)
    ok, rep = bf16_close(out.view(M, Hq + 2 * Hkv, D)[:, :Hq], qr, ulps=2)
    assert ok, rep
    kp = kpool.cpu().permute(0, 1, 3, 2, 4).reshape(n_pages, Hkv, 64, D)
    vp = vpool.cpu()[..., VSLOT].permute(0, 1, 3, 2)
    for m in range(M):
        page, within = int(bt[m, int(slot[m]) // 64]), int(slot[m]) % 64
        ok, rep = bf16_close(kp[page, :, within], kr[m], ulps=2)
        assert ok, (m, rep)
        ok, rep = bf16_close(vp[page, :, within], qkv[m, Hq + Hkv:], ulps=2)
        assert ok, (m, rep)
    assert int((kpool != 0).sum().cpu()) <= M * Hkv * D and int((vpool != 0).sum().cpu()) <= M * Hkv * D


_TRIPS_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from mlx_vlm_amd import ops as vops
BF = torch.bfloat16
def rnd(*shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(BF)
out = {}
for M in (9, 16):
    for K, I in ((1536, 1024), (3584, 512), (896, 256)):
        h = rnd(M, K, seed=M + K).cuda()
        nw = (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(5))).to(BF).cuda()
        w = rnd(2 * I, K, seed=K, scale=0.05).cuda()
        out[f"swiglu_{M}_{K}"] = vops.gemv_ws(h, w, norm_w=nw, eps=1e-6, epilogue=vops.EPI_SWIGLU).cpu()
        out[f"plain_{M}_{K}"] = vops.gemv_ws(h, w).cpu()
torch.save(out, sys.argv[2])
"""


def test_gemv_mfma_row_trip_prologue_is_bit_identical_to_the_loop_form(vops, tmp_path):
    """The activation prologue only changes HOW the rows reach LDS (all loads of a trip in flight vs one dependent round
    trip per chunk): same sum-of-squares order, same typed normalisation -> the outputs are the same bits.  The loop form
    is selected by an environment knob read once per process, so it runs in a child process."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for knob in ("0", "1"):
        path = str(tmp_path / f"trips{knob}.pt")
        env = dict(os.environ, VLM_GEMV_MFMA_TRIPS=knob, VLM_GEMV_MFMA2="0")      # the first form (the second has no prologue of its own)
        r = subprocess.run([sys.executable, "-c", _TRIPS_CHILD, root, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[knob] = torch.load(path)
    assert res["0"].keys() == res["1"].keys() and len(res["0"]) == 12
    for k in res["0"]:
        assert torch.equal(res["0"][k], res["1"][k]), k


# ------------------------------------------------------------------ batched decode rows over 4-bit weights (gemv_mfma.hip, W4 form)
@pytest.mark.parametrize("M", [5, 8, 16])
@pytest.mark.parametrize("N,K", [(2048, 1536), (1536, 8960), (302, 256), (1024, 3584)])
def test_gemv_w4_mfma_rows_plain_bias_residual(vops, M, N, K):
    """vlm_gemv_w4_ws for 5..16 rows vs nn.QuantizedLinear restated: nibbles as bf16 128 + q in the MFMA A fragments, every
    64-wide group scaled in fp32 (exact affine form), bias as a second typed op; split-K forms through the workspace;
    repeatable bit for bit."""
    ow, dw = _q4(N, K, seed=310 + M)
    x, b, r = rnd(M, K, seed=311), rnd(N, seed=312, scale=0.3), rnd(M, N, seed=313)
    out = vops.gemv_w4_ws(x.cuda(), dw.wq, dw.sb)
    ok, rep = bf16_close(out, ow.linear(x), ulps=2)
    assert ok, rep
    assert torch.equal(vops.gemv_w4_ws(x.cuda(), dw.wq, dw.sb), out)
    ok, rep = bf16_close(vops.gemv_w4_ws(x.cuda(), dw.wq, dw.sb, bias=b.cuda(), epilogue=vops.EPI_BIAS), ow.linear(x, b), ulps=2)
    assert ok, rep
    rr = r.cuda().clone()
    out2 = vops.gemv_w4_ws(x.cuda(), dw.wq, dw.sb, res=rr, out=rr, epilogue=vops.EPI_RESIDUAL)
    # r + lin cancels: the error of the sum is one rounding step of the INTERMEDIATE lin (rms 2.2 here: an ulp of 2^-6 where
    # the sum can be near 0) plus the sum's own rounding, so the bound is stated on |lin| + |sum|
    lin = ow.linear(x).float()
    ref2 = O.add(r, ow.linear(x)).float()
    err = (out2.float().cpu() - ref2).abs()
    tol = 2.0 ** -7 * lin.abs() + 2 * 2.0 ** -7 * ref2.abs() + 2e-3 * float(ref2.pow(2).mean().sqrt())
    assert bool((err <= tol).all()), (float(err.max()), int((err > tol).sum()))


@pytest.mark.parametrize("M", [8, 16])
def test_gemv_w4_mfma_rows_norm_swiglu_head_and_qkv(vops, M):
    from mlx_vlm_amd.models import quantized as Qz

    K, I = 1536, 2048
    h, nw = rnd(M, K, seed=320), (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(321))).to(BF)
    og, dg = _q4(I, K, seed=322)
    ou, du = _q4(I, K, seed=323)
    dgu = Qz.interleave_rows(dg, du)
    xn = O.rms_norm(h, nw, 1e-6)
    out = vops.gemv_w4_ws(h.cuda(), dgu.wq, dgu.sb, norm_w=nw.cuda(), eps=1e-6, epilogue=vops.EPI_SWIGLU)
    ok, rep = bf16_close(out, O.swiglu(og.linear(xn), ou.linear(xn)), ulps=3)
    assert ok, rep
    oh, dh = _q4(9000, K, seed=324)
    ok, rep = bf16_close(vops.gemv_w4_ws(h.cuda(), dh.wq, dh.sb, norm_w=nw.cuda(), eps=1e-6), oh.linear(xn), ulps=2)
    assert ok, rep
    if M < 9:
        return
    # qkv + M-RoPE + paged KV write over 4-bit q / k / v rows
    Hq, Hkv, D = 12, 2, 128
    parts = [_q4(Hq * D, K, seed=330), _q4(Hkv * D, K, seed=331), _q4(Hkv * D, K, seed=332)]
    dq = Qz.cat_rows([p[1] for p in parts])
    bqkv = rnd((Hq + 2 * Hkv) * D, seed=333, scale=0.3)
    g = torch.Generator().manual_seed(334)
    pos = torch.randint(0, 3000, (M,), generator=g, dtype=torch.int32)
    max_pages = 4
    slot = torch.randint(0, 64 * max_pages, (M,), generator=g, dtype=torch.int32)
    inv = O.mrope_inv_freq(D, 1e6)
    sel = O.chunked_position_selector([16, 24, 24], D // 2)
    bs = [bqkv[:Hq * D], bqkv[Hq * D:(Hq + Hkv) * D], bqkv[(Hq + Hkv) * D:]]
    qkv = torch.cat([p[0].linear(xn, b) for p, b in zip(parts, bs)], -1).view(M, Hq + 2 * Hkv, D)
    p3 = pos.long()[None, :, None].expand(3, M, 1)
    qr = O.mrope_apply(qkv[:, :Hq][:, :, None], p3, inv, sel, "fused")[:, :, 0]
    kr = O.mrope_apply(qkv[:, Hq:Hq + Hkv][:, :, None], p3, inv, sel, "fused")[:, :, 0]
    n_pages = M * max_pages
    bt = torch.randperm(n_pages, generator=g).to(torch.int32).reshape(M, max_pages)
    kpool = torch.zeros(n_pages, Hkv, D // 8, 64, 8, dtype=BF, device="cuda")
    vpool = torch.zeros(n_pages, Hkv, D, 64, dtype=BF, device="cuda")
    out = vops.gemv_w4_qkv_rope_kvwrite_ws(h.cuda(), nw.cuda(), dq.wq, dq.sb, bqkv.cuda(), Hq, Hkv, D, pos.cuda(), slot.cuda(),
                                           inv.cuda(), bt.cuda(), kpool, vpool)
    ok, rep = bf16_close(out.view(M, Hq + 2 * Hkv, D)[:, :Hq], qr, ulps=2)
    assert ok, rep
    kp = kpool.cpu().permute(0, 1, 3, 2, 4).reshape(n_pages, Hkv, 64, D)
    vp = vpool.cpu()[..., VSLOT].permute(0, 1, 3, 2)
    for m in range(M):
        page, within = int(bt[m, int(slot[m]) // 64]), int(slot[m]) % 64
        ok, rep = bf16_close(kp[page, :, within], kr[m], ulps=2)
        assert ok, (m, rep)
        ok, rep = bf16_close(vp[page, :, within], qkv[m, Hq + Hkv:], ulps=2)
        assert ok, (m, rep)
