"""SURVEY section 8(b) cache contract, host side: the bookkeeping of the paged KVCache / BatchKVCache facades against the
reference's own classes run over the shim (tests/golden/make_golden_batchcache.py -> batchcache_ref.npz).  The device write of
update_and_fetch is replaced by its bookkeeping (no GPU here); tests/test_cache_contract_gpu.py replays the same scenario with the
real kernel and compares the contents."""
import numpy as np
import pytest
import torch

from tests.cache_contract_replay import replay


def _dry_append(self, keys, values):
    s = self._seq
    S = int(keys.shape[2])
    if S:
        s.reserve(self.offset + S)
        s.advance_layer(self._layer, S)


def _dry_update(self, keys, values):
    _dry_append(self, keys, values)
    return self.state


@pytest.mark.parametrize("layout", ["paged", "identity"])
def test_cache_facades_bookkeeping_matches_reference_classes(monkeypatch, layout):
    from mlx_vlm_amd.models import cache as C

    monkeypatch.setattr(C.KVCache, "update_and_fetch", _dry_update)
    monkeypatch.setattr(C.KVCache, "_append", _dry_append)          # (BatchKVCache appends row by row without materialising the state)
    pool = C.KVPool(2, 2, 128, max_tokens=1024, max_seqs=32, max_pages_per_seq=4, device="cpu", layout=layout)
    _, n_ops = replay(pool, "cpu", check_contents=False)
    assert n_ops >= 35


def test_update_and_fetch_refuses_host_tensors_and_bad_shapes():
    from mlx_vlm_amd.models import cache as C

    pool = C.KVPool(2, 2, 128, max_tokens=256, max_seqs=4, device="cpu", layout="paged")
    c = C.KVCache(C.PagedSequence(pool), 0)
    k = torch.zeros(1, 2, 3, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="device tensors"):
        c.update_and_fetch(k, k)
    with pytest.raises(ValueError):
        c.update_and_fetch(k[:, :1], k[:, :1])
    with pytest.raises(IndexError):
        c.extract(2)
    assert c.extract(-1)._seq is c._seq


def test_batch_cache_left_padding_only_on_an_empty_cache():
    from mlx_vlm_amd.models import cache as C

    pool = C.KVPool(1, 2, 128, max_tokens=256, max_seqs=4, device="cpu", layout="paged")
    b = C.BatchKVCache.for_layers(pool, [0, 0])[0]
    b._row(0).set_offset(3)
    with pytest.raises(ValueError, match="Left padding can only be added to an empty BatchKVCache"):       # the reference's words
        b.prepare(left_padding=[1, 0])
    assert b.is_trimmable() and b.make_mask(1) is None and b.make_mask(4) == "causal" and not b.is_single_row()
