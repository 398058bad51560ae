"""Pin the oracle (oracle/) against the committed HuggingFace goldens
(tests/golden/qwen2_vl_tiny_hf.npz, made by tests/golden/make_golden.py) and
against the reference's own self-consistency contracts.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import image_processor as ip
from oracle import ops
from oracle import qwen2_vl as oq

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen2_vl_tiny_hf.npz"))


@pytest.fixture(scope="module")
def model():
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=torch.float32, std=0.05, embed_std=0.2)
    return cfg, W


@pytest.mark.parametrize("case", ["one_image", "two_images"])
def test_image_processor_matches_hf_inputs(case):
    sizes = G[case + ".sizes"]
    flat = G[case + ".images"]
    imgs, off = [], 0
    for h, w in sizes.tolist():
        imgs.append(flat[off:off + 3 * h * w].reshape(3, h, w))
        off += 3 * h * w
    pix, thw = ip.process(imgs)
    np.testing.assert_array_equal(thw, G[case + ".grid_thw"])
    np.testing.assert_array_equal(pix, G[case + ".pixel_values"])


def test_smart_resize_table():
    for h, w, rh, rw in G["smart_resize.table"].tolist():
        assert ip.smart_resize(h, w) == (rh, rw)


@pytest.mark.parametrize("tag", ["ip_a", "ip_b"])
def test_patchify_matches_hf_pil_processor(tag):
    """processing_qwen3_vl.py:302-354 vs HF Qwen2VLImageProcessorPil: same
    bicubic resize, rescale, normalise and 10-D transpose."""
    im = G[tag + ".image_hwc"]
    pix, thw = ip.process_one(np.transpose(im, (2, 0, 1)))
    np.testing.assert_array_equal(np.array([thw]), G[tag + ".hf_grid_thw"])
    np.testing.assert_allclose(pix.astype(np.float64).sum(axis=1), G[tag + ".hf_pixel_values_rowsum"], rtol=0, atol=2e-3)
    full = G[tag + ".hf_pixel_values"]
    if full.shape[0]:
        np.testing.assert_allclose(pix, full, rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", ["one_image", "two_images"])
def test_rope_index_matches_hf(case, model):
    cfg, _ = model
    pos, deltas = oq.get_rope_index(cfg, G[case + ".input_ids"], G[case + ".grid_thw"])
    np.testing.assert_array_equal(pos, G[case + ".hf_position_ids"])
    np.testing.assert_array_equal(deltas, G[case + ".hf_rope_deltas"])


def test_rope_index_text_only_padded_matches_hf(model):
    if "text_padded.input_ids" not in G.files:
        pytest.skip("HF text-only rope index golden not available")
    cfg, _ = model
    pos, deltas = oq.get_rope_index(cfg, G["text_padded.input_ids"], attention_mask=G["text_padded.attention_mask"])
    hp = G["text_padded.hf_position_ids"]
    if hp.ndim == 3:  # HF returns [3,B,L]; the reference's text-only branch returns [B,L]
        assert (hp[0] == hp[1]).all() and (hp[0] == hp[2]).all()
        hp = hp[0]
    am = G["text_padded.attention_mask"].astype(bool)
    # padded slots: the reference writes 1 (language.py:386-388), HF 5.x writes 0 - never attended
    np.testing.assert_array_equal(pos[am], hp[am])
    assert (pos[~am] == 1).all()
    # deltas: the reference keeps max_pos + 1 - padded_len (language.py:389-390), i.e. next position =
    # padded_len + delta = number of valid tokens; HF 5.x reports the delta against the valid length (0).
    L = am.shape[1]
    np.testing.assert_array_equal(deltas[:, 0] + L, am.sum(-1))
    np.testing.assert_array_equal(G["text_padded.hf_rope_deltas"][:, 0] + am.sum(-1), am.sum(-1))


@pytest.mark.parametrize("case", ["one_image", "two_images"])
def test_vision_tower_matches_hf(case, model):
    cfg, W = model
    feats = oq.vision_tower(W, cfg, torch.from_numpy(G[case + ".pixel_values"]), G[case + ".grid_thw"])
    np.testing.assert_allclose(feats.numpy(), G[case + ".hf_image_features"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", ["one_image", "two_images"])
def test_full_forward_logits_match_hf(case, model):
    cfg, W = model
    ids = G[case + ".input_ids"]
    emb, pos, _ = oq.get_input_embeddings(W, cfg, ids, torch.from_numpy(G[case + ".pixel_values"]), G[case + ".grid_thw"])
    h = oq.qwen2_model(W, cfg, emb, None, torch.from_numpy(pos))
    logits = oq.lm_head(W, cfg, h)[0].numpy()
    np.testing.assert_allclose(logits, G[case + ".hf_logits"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("case", ["one_image", "two_images"])
def test_greedy_generate_matches_hf(case, model):
    """generate_step greedy (ar.py:151-515) with KV cache + rope deltas vs HF generate."""
    cfg, W = model
    toks = oq.generate_greedy(W, cfg, G[case + ".input_ids"], torch.from_numpy(G[case + ".pixel_values"]),
                              G[case + ".grid_thw"], max_tokens=8)
    assert toks == G[case + ".hf_greedy"].tolist()


def test_mrope_fused_equals_fallback_fp32():
    """The reference's own contract (tests/test_rope_utils.py:366-407): fused
    kernel == pure fallback within 1e-4 in fp32, 2-D and 3-D position ids."""
    g = torch.Generator().manual_seed(0)
    q = torch.randn(2, 3, 7, 128, generator=g)
    inv = ops.mrope_inv_freq(128, 1e6)
    sel = ops.chunked_position_selector([16, 24, 24], 64)
    pos3 = torch.randint(0, 500, (3, 2, 7), generator=g)
    a = ops.mrope_apply(q, pos3, inv, sel, "fused")
    b = ops.mrope_apply(q, pos3, inv, sel, "fallback")
    assert torch.allclose(a, b, atol=1e-4)
    pos2 = torch.randint(0, 500, (2, 7), generator=g)
    a = ops.mrope_apply(q, pos2, inv, sel, "fused")
    b = ops.mrope_apply(q, pos2, inv, sel, "fallback")
    assert torch.allclose(a, b, atol=1e-4)
    # text-only: 3 equal axes == scalar positions
    a3 = ops.mrope_apply(q, pos2[None].expand(3, -1, -1), inv, sel, "fused")
    assert torch.equal(a, a3)


def test_selector_chunked():
    sel = ops.chunked_position_selector([16, 24, 24], 64).tolist()
    assert sel == [0] * 16 + [1] * 24 + [2] * 24


def test_kv_cache_growth_and_trim():
    c = ops.KVCache()
    k = torch.arange(2 * 5 * 4, dtype=torch.float32).reshape(1, 2, 5, 4)
    ks, vs = c.update_and_fetch(k, k + 1)
    assert ks.shape == (1, 2, 5, 4) and c.keys.shape[2] == 256 and c.offset == 5
    big = torch.ones(1, 2, 300, 4)
    ks, _ = c.update_and_fetch(big, big)
    assert c.offset == 305 and ks.shape[2] == 305 and torch.equal(ks[:, :, :5], k)
    assert c.trim(5) == 5 and c.offset == 300


def test_sampler_filters_small_known_answers():
    lp = torch.log(torch.tensor([[0.1, 0.4, 0.2, 0.3]]))
    assert ops.argmax_first(lp).tolist() == [1]
    k2 = ops.apply_top_k(lp, 2)
    assert torch.isinf(k2[0, 0]) and torch.isinf(k2[0, 2]) and not torch.isinf(k2[0, 1])
    # top_p 0.6: ascending cum = .1,.3,.6,1.0 ; keep cum > 0.4 -> probs .3,.4
    p = ops.apply_top_p(lp, 0.6)
    assert torch.isinf(p[0, 0]) and torch.isinf(p[0, 2]) and not torch.isinf(p[0, 3]) and not torch.isinf(p[0, 1])
    m = ops.apply_min_p(lp, 0.6)  # threshold .24
    assert torch.isinf(m[0, 0]) and torch.isinf(m[0, 2]) and not torch.isinf(m[0, 3])
    # ties resolve to the lowest index (MLX argmax semantics)
    assert ops.argmax_first(torch.tensor([[1.0, 3.0, 3.0]])).tolist() == [1]


def test_categorical_gumbel_distribution():
    lp = torch.log(torch.tensor([0.1, 0.6, 0.3]))
    cnt = np.zeros(3)
    for step in range(3000):
        cnt[ops.categorical_gumbel(lp, 1.0, seed=3, step=step)] += 1
    np.testing.assert_allclose(cnt / cnt.sum(), [0.1, 0.6, 0.3], atol=0.03)
