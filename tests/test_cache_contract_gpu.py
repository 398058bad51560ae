"""SURVEY section 8(b) cache contract on the GPU: KVCache.update_and_fetch / state / trim / extract and BatchKVCache.update_and_fetch /
prepare / finalize / filter / extend / extract / merge / trim through the facades over the paged pool (vlm_kv_append_tokens writes the
rows), every operation's bookkeeping AND contents equal to the reference's own classes run over the shim."""
import pytest
import torch

from tests.cache_contract_replay import replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout", ["paged", "identity"])
def test_cache_facades_match_reference_classes(layout):
    from mlx_vlm_amd.models import cache as C

    pool = C.KVPool(2, 2, 128, max_tokens=1024, max_seqs=32, max_pages_per_seq=4, device="cuda", layout=layout)
    n_checked, n_ops = replay(pool, "cuda", check_contents=True)
    assert n_ops >= 35 and n_checked > 100


def test_update_and_fetch_then_engine_attention_sees_the_rows():
    """rows written through the facade are exactly where the decode attention kernel looks: update_and_fetch K / V, then
    vlm_attn_decode_paged over the same pool against plain softmax(q k^T) v"""
    from mlx_vlm_amd import ops
    from mlx_vlm_amd.models import cache as C

    torch.manual_seed(3)
    Hkv, Hq, D, S = 2, 4, 128, 70           # crosses a page boundary
    pool = C.KVPool(1, Hkv, D, max_tokens=512, max_seqs=4, max_pages_per_seq=4, device="cuda", layout="paged")
    c = C.KVCache(C.PagedSequence(pool), 0)
    k = (torch.randn(1, Hkv, S, D, device="cuda") * 0.5).to(torch.bfloat16)
    v = (torch.randn(1, Hkv, S, D, device="cuda") * 0.5).to(torch.bfloat16)
    rk, rv = c.update_and_fetch(k[:, :, :40], v[:, :, :40])
    assert torch.equal(rk, k[:, :, :40])
    rk, rv = c.update_and_fetch(k[:, :, 40:].transpose(1, 2).contiguous().transpose(1, 2), v[:, :, 40:])      # a strided view
    assert torch.equal(rk, k) and torch.equal(rv, v) and c.offset == S and c.nbytes > 0
    q = (torch.randn(1, Hq * D, device="cuda") * 0.5).to(torch.bfloat16)
    kv_len = torch.tensor([S], dtype=torch.int32, device="cuda")
    kp, vp = pool.kpool[0], pool.vpool[0]
    out = ops.attn_decode_paged(q, kp, vp, pool.block_table[c._seq.seq:], kv_len, 0, Hq, Hkv, D, D ** -0.5, 1, max_pages=pool.max_pages)
    qf = q.view(Hq, D).float()
    ref = torch.empty(Hq, D)
    for h in range(Hq):
        g = h // (Hq // Hkv)
        p = torch.softmax((k[0, g].float() @ qf[h]) * D ** -0.5, dim=0)
        ref[h] = (p[:, None] * v[0, g].float()).sum(0).cpu()
    assert (out.view(Hq, D).float().cpu() - ref).abs().max() < 2e-2
