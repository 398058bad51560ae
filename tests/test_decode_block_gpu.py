"""GPU parity tests of the fused decode block (csrc/decode_block.hip, vlm_decode_block_bf16): ONE launch for
[page-split attention] [split merge + o_proj + residual] [RMSNorm + gate/up + SwiGLU] of a one-row decode step.

Two bars:
  * BIT-IDENTICAL to the three launches it replaces (vlm_attn_decode_paged_split partial form -> vlm_gemv_attn_out_bf16
    -> vlm_gemv_bf16 with the norm prologue and the SwiGLU epilogue) - same source expressions, same summation orders;
  * the oracle (oracle/ops.py: SDPA -> Linear -> residual -> RMSNorm -> SwiGLU MLP front half) within the tolerance of
    the unfused path's own tests (test_ops_gpu.py: P and the partials are rounded to bf16).
The hand-offs inside the launch (sc1 granules tagged with a launch epoch) are exercised over many back-to-back
launches on the same workspace, on changing inputs, with every output word checked (MI355X_MICROARCH.md: "test every
hand-off under uneven load, consumer L1-warm, checking every word"), and beside a concurrent streaming kernel.
"""
import pytest
import torch

from oracle import ops as O
from tests.helpers import bf16_close
from tests.test_ops_gpu import VSLOT, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
D = 128


@pytest.fixture(scope="module")
def vops():
    from mlx_vlm_amd import ops

    return ops


def _pools(n, Hkv, seed, identity=True, extra_pages=2):
    max_pages = (n + 63) // 64 + extra_pages
    kpool = torch.full((max_pages, Hkv, D // 8, 64, 8), float("nan"), dtype=BF)
    vpool = torch.full((max_pages, Hkv, D, 64), float("nan"), dtype=BF)
    k, v = rnd(n, Hkv, D, seed=seed), rnd(n, Hkv, D, seed=seed + 1)
    order = list(range(max_pages))
    bt = None
    if not identity:
        g = torch.Generator().manual_seed(seed + 2)
        order = torch.randperm(max_pages, generator=g).tolist()
        bt = torch.tensor([order], dtype=torch.int32)
    for p in range((n + 63) // 64):
        m = min(64, n - p * 64)
        kpool[order[p], :, :, :m, :] = k[p * 64:p * 64 + m].permute(1, 0, 2).reshape(Hkv, m, D // 8, 8).permute(0, 2, 1, 3)
        vpool[order[p]][:, :, VSLOT[:m]] = v[p * 64:p * 64 + m].permute(1, 2, 0)
    return k, v, kpool, vpool, bt, max_pages


def _unfused(vops, q, kpool, vpool, bt, n, Hq, Hkv, scale, nsplit, wo, h, ln2, eps, wgu, max_pages):
    from mlx_vlm_amd import _lib
    kv_len = torch.tensor([n], dtype=torch.int32).cuda()
    po, pml = vops.attn_decode_paged_split(q, kpool, vpool, bt, kv_len, 0, Hq, Hkv, D, scale, nsplit, max_pages=max_pages, merge=False)
    h2 = vops.gemv_attn_out_bf16_(po, pml, wo, h.clone(), Hq, D)
    act = vops.gemv(h2, wgu, norm_w=ln2, eps=eps, epilogue=vops.EPI_SWIGLU)
    return h2, act


def _skip_unless_supported(vops, Hq, Hkv, inter, nsplit):
    if not vops.decode_block_supported(Hq, Hkv, D, inter, nsplit):
        pytest.skip("the fused decode block does not take this shape on this device")


@pytest.mark.parametrize("n,heads,nsplit,inter,identity", [
    (386, (12, 2), 16, 8960, True), (1, (12, 2), 16, 8960, True), (64, (12, 2), 16, 8960, False),
    (1024, (12, 2), 16, 8960, True), (1500, (12, 2), 16, 8960, False), (700, (12, 4), 8, 8960, True),
    (130, (12, 4), 8, 4096, False), (450, (12, 2), 16, 8954, True), (2000, (12, 2), 8, 8960, False)])
def test_decode_block_bit_identical_to_three_launches_and_close_to_oracle(vops, n, heads, nsplit, inter, identity):
    Hq, Hkv = heads
    _skip_unless_supported(vops, Hq, Hkv, inter, nsplit)
    K = Hq * D
    scale, eps = D ** -0.5, 1e-6
    k, v, kpool, vpool, bt, max_pages = _pools(n, Hkv, seed=300 + n, identity=identity)
    q = rnd(1, K, seed=61)
    wo = rnd(K, K, seed=64, scale=0.03)
    h = rnd(1, K, seed=65)
    ln2 = (1.0 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(66))).to(BF)
    wgu = rnd(2 * inter, K, seed=67, scale=0.03)
    dq, dk, dv, dbt = q.cuda(), kpool.cuda(), vpool.cuda(), (bt.cuda() if bt is not None else None)
    dwo, dln, dwgu = wo.cuda(), ln2.cuda(), wgu.cuda()
    h_ref, act_ref = _unfused(vops, dq, dk, dv, dbt, n, Hq, Hkv, scale, nsplit, dwo, h.cuda(), dln, eps, dwgu, max_pages)
    kv_len = torch.tensor([n], dtype=torch.int32).cuda()
    hf = h.cuda().clone()
    act = torch.full((1, inter), float("nan"), dtype=BF, device="cuda")
    hf, act, ws = vops.decode_block_(dq, dk, dv, dbt, kv_len, 0, Hq, Hkv, D, scale, nsplit, dwo, hf, dln, eps, dwgu, act,
                                     max_pages=max_pages)
    torch.cuda.synchronize()
    err, _ = vops.decode_block_debug(ws)
    assert err == 0, f"{err} hand-offs gave up"
    assert torch.equal(hf.view(torch.int16), h_ref.view(torch.int16)), "residual stream differs from the three launches"
    assert torch.equal(act.view(torch.int16), act_ref.view(torch.int16)), "gate/up output differs from the three launches"
    # the oracle: SDPA -> o_proj + residual -> RMSNorm -> swiglu(gate, up)
    att = O.sdpa(q.view(1, Hq, 1, D), k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None], scale)[0, :, 0].reshape(1, K)
    h_o = (O.linear(att, wo).float() + h.float()).to(BF)
    ok, rep = bf16_close(hf, h_o, ulps=3, atol_rms=2e-2)
    assert ok, rep
    xn = O.rms_norm(hf.cpu(), ln2, eps)                 # from the kernel's own h: isolates the MLP half
    gu = O.linear(xn, wgu)
    act_o = O.swiglu(gu[:, 0::2], gu[:, 1::2])
    ok, rep = bf16_close(act, act_o, ulps=2, atol_rms=2e-2)
    assert ok, rep


def test_decode_block_many_launches_changing_inputs_every_word(vops):
    """60 back-to-back launches on ONE workspace (the epoch advances per launch, granule buffers are reused), the query and
    the residual stream change every launch, no host synchronisation in between: a stale granule or a flag seen early shows
    up as a difference from the three-launch path, which runs afterwards on the same inputs."""
    Hq, Hkv, nsplit, inter, n = 12, 2, 16, 8960, 777
    _skip_unless_supported(vops, Hq, Hkv, inter, nsplit)
    K = Hq * D
    scale, eps = D ** -0.5, 1e-6
    _, _, kpool, vpool, _, max_pages = _pools(n, Hkv, seed=900)
    dk, dv = kpool.cuda(), vpool.cuda()
    wo, ln2, wgu = rnd(K, K, seed=1, scale=0.03).cuda(), rnd(K, seed=2, scale=0.2).cuda() + 1, rnd(2 * inter, K, seed=3, scale=0.03).cuda()
    ln2 = ln2.to(BF)
    kv_len = torch.tensor([n], dtype=torch.int32).cuda()
    R = 60
    qs = [rnd(1, K, seed=1000 + i).cuda() for i in range(R)]
    hs = [rnd(1, K, seed=2000 + i).cuda() for i in range(R)]
    outs_h = [x.clone() for x in hs]
    outs_a = [torch.empty(1, inter, dtype=BF, device="cuda") for _ in range(R)]
    ws = None
    torch.cuda.synchronize()
    for i in range(R):
        _, _, ws = vops.decode_block_(qs[i], dk, dv, None, kv_len, 0, Hq, Hkv, D, scale, nsplit, wo, outs_h[i], ln2, eps, wgu,
                                      outs_a[i], ws=ws, max_pages=max_pages)
    torch.cuda.synchronize()
    err, _ = vops.decode_block_debug(ws)
    assert err == 0
    for i in range(R):
        h_ref, act_ref = _unfused(vops, qs[i], dk, dv, None, n, Hq, Hkv, scale, nsplit, wo, hs[i], ln2, eps, wgu, max_pages)
        assert torch.equal(outs_h[i].view(torch.int16), h_ref.view(torch.int16)), f"launch {i}: residual stream"
        assert torch.equal(outs_a[i].view(torch.int16), act_ref.view(torch.int16)), f"launch {i}: gate/up output"


def test_decode_block_beside_a_streaming_kernel(vops):
    """uneven load: a large copy runs on a second stream while the block's launches run (the hand-offs then complete under
    a busy memory system and a perturbed dispatch) - results still bit-identical, no hand-off gives up."""
    Hq, Hkv, nsplit, inter, n = 12, 2, 16, 8960, 450
    _skip_unless_supported(vops, Hq, Hkv, inter, nsplit)
    K = Hq * D
    scale, eps = D ** -0.5, 1e-6
    _, _, kpool, vpool, _, max_pages = _pools(n, Hkv, seed=901)
    dk, dv = kpool.cuda(), vpool.cuda()
    wo, wgu = rnd(K, K, seed=1, scale=0.03).cuda(), rnd(2 * inter, K, seed=3, scale=0.03).cuda()
    ln2 = torch.ones(K, dtype=BF, device="cuda")
    kv_len = torch.tensor([n], dtype=torch.int32).cuda()
    q, h = rnd(1, K, seed=5).cuda(), rnd(1, K, seed=6).cuda()
    h_ref, act_ref = _unfused(vops, q, dk, dv, None, n, Hq, Hkv, scale, nsplit, wo, h, ln2, eps, wgu, max_pages)
    big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    ws = None
    torch.cuda.synchronize()
    for rep in range(8):
        with torch.cuda.stream(side):
            big2 = big.clone()
        for i in range(5):
            hf = h.clone()
            act = torch.empty(1, inter, dtype=BF, device="cuda")
            _, _, ws = vops.decode_block_(q, dk, dv, None, kv_len, 0, Hq, Hkv, D, scale, nsplit, wo, hf, ln2, eps, wgu, act, ws=ws,
                                          max_pages=max_pages)
            assert torch.equal(hf.view(torch.int16), h_ref.view(torch.int16)), (rep, i)
            assert torch.equal(act.view(torch.int16), act_ref.view(torch.int16)), (rep, i)
        del big2
    torch.cuda.synchronize()
    err, _ = vops.decode_block_debug(ws)
    assert err == 0
