"""Idefics2: the oracle against vectors produced by the REFERENCE'S OWN files
(tests/golden/make_golden_ref_idefics2.py ran mlx_vlm/models/idefics2/*.py, models/base.py, models/cache.py and
generate/ar.py unmodified over oracle/mlx_shim) and against transformers' own Idefics2 image processor (PIL backend, run
by the same script); only the .npz is read here.  CPU only."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import idefics2 as oi
from oracle import ops

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "idefics2_tiny_ref.npz"))
ROWS = slice(None, None, 3)
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def _weights(dt):
    W = oi.random_weights(oi.tiny_cfg(), seed=4321, dtype=torch.float32, **oi.TEST_WEIGHT_SCALES)
    return {k: v.to(dt) for k, v in W.items()}


def _from_bits(a):
    return torch.from_numpy((a.astype(np.uint32) << 16).view(np.float32).copy()).to(torch.bfloat16)


def _close(name, got, ref, dt, rtol=2e-4, atol=2e-5):
    got = got.detach().to(torch.float32).numpy()
    if dt == "f32":
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=name)
    else:
        assert np.array_equal(got, ref), (name, int((got != ref).sum()), float(np.abs(got - ref).max()))


@pytest.mark.parametrize("name,kw", [("default", {}), ("split", {"do_image_splitting": True}),
                                    ("small", {"shortest_edge": 56, "longest_edge": 140})])
def test_image_processor_bit_exact_vs_transformers(name, kw):
    """resize rule (shortest_edge / longest_edge), bilinear resize, rescale + normalise, padding to the batch maximum, the
    pixel attention mask, padding images for samples with fewer images, 4 + 1 image splitting: crc-exact."""
    which = [[int(i) for i in row if i >= 0] for row in G[f"proc.{name}.which"]]
    pv, pm = oi.preprocess([[G[f"img{i}.image_hwc"] for i in row] for row in which], **kw)
    assert list(pv.shape) == G[f"proc.{name}.pixel_shape"].tolist() and pv.dtype == np.float32
    np.testing.assert_allclose(pv.astype(np.float64).sum(axis=(2, 3, 4)), G[f"proc.{name}.pixel_sum"], rtol=0, atol=1e-6)
    assert zlib.crc32(np.ascontiguousarray(pv).tobytes()) == int(G[f"proc.{name}.pixel_crc32"][0])
    assert np.array_equal(pm.sum(axis=(2, 3)), G[f"proc.{name}.mask_sum"])
    assert zlib.crc32(np.ascontiguousarray(pm.astype(np.int64)).tobytes()) == int(G[f"proc.{name}.mask_crc32"][0])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("ci", [0, 1])
def test_tower_connector_and_scatter_vs_reference(ci, dt):
    """Padding-image removal + patch mask, bucketed position ids, the unmasked encoder, post_layernorm, modality projection,
    perceiver resampler (GQA cross-attention over [context | latents]), masked_scatter.  bf16: from the reference's own
    patch embeddings on (Conv2d vs GEMM summation order) everything is bit-exact."""
    cfg, W = oi.tiny_cfg(), _weights(DT[dt])
    p = f"case{ci}.{dt}."
    pv = torch.from_numpy(G[f"case{ci}.pixel_values"]).to(DT[dt])
    pm = G[f"case{ci}.pixel_attention_mask"]
    real, pmask = oi.real_images_and_patch_mask(pv, pm, cfg.vision.patch_size)
    assert np.array_equal(pmask, G[f"case{ci}.patch_mask"])
    start = None
    if dt == "bf16":
        ref = _from_bits(G[p + "ref_vision_embeddings_bits"])
        own = oi.vision_embeddings(W, cfg, real, pmask)
        d = (own.float() - ref.float()).abs()
        assert float(d.max()) <= 2.0 ** -6 * float(ref.float().abs().max()) and float((d > 0).float().mean()) < 5e-3
        start = ref
    pooled, states = oi.vision_tower(W, cfg, real, pmask, embeddings=start, return_states=True)
    _close("embeddings", states[0][:, ROWS], G[p + "ref_vision_embeddings"], dt)
    _close("layer0", states[1][:, ROWS], G[p + "ref_vision_layer0"], dt)
    _close("pooled", pooled[:, ROWS], G[p + "ref_pooled"], dt)
    feats = oi.connector(W, cfg, pooled)
    _close("image_features", feats, G[p + "ref_image_features"], dt, rtol=1e-3, atol=1e-4)
    emb = oi.get_input_embeddings(W, cfg, G[f"case{ci}.input_ids"], pv, pm, vision_embeddings_override=start)
    _close("inputs_embeds", emb[0, ROWS], G[p + "ref_inputs_embeds"], dt, rtol=1e-3, atol=1e-4)
    if dt == "bf16":
        assert torch.equal(emb[0], _from_bits(G[p + "ref_inputs_embeds_bits"]))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("ci", [0, 1])
def test_prefill_and_kvcache_decode_vs_reference(ci, dt):
    """Mistral decoder from the reference's own spliced prompt: bias-free projections, nn.RoPE at cache offsets, GQA, KVCache
    growth, untied head; 6 greedy steps - bf16 bit for bit."""
    cfg, W = oi.tiny_cfg(), _weights(DT[dt])
    p = f"case{ci}.{dt}."
    if dt == "bf16":
        emb = _from_bits(G[p + "ref_inputs_embeds_bits"])[None]
    else:
        emb = oi.get_input_embeddings(W, cfg, G[f"case{ci}.input_ids"], torch.from_numpy(G[f"case{ci}.pixel_values"]),
                                      G[f"case{ci}.pixel_attention_mask"])
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    logits = oi.language_model(W, cfg, emb, cache, last_only=True)[:, -1, :]
    tol = dict(rtol=2e-3, atol=2e-4)
    _close("prefill last", logits[0], G[p + "ref_prefill_logits_last"], dt, **tol)
    toks, rows = [], []
    for n in range(7):
        y = int(ops.argmax_first(ops.logprobs_from_logits(logits))[0])
        toks.append(y)
        rows.append(logits[0].clone())
        logits = oi.language_model(W, cfg, oi.embed_tokens(W, np.array([[y]])), cache)[:, -1, :]
    ref_t = G[p + "ref_greedy"].tolist()
    if dt == "bf16":
        assert toks[:6] == ref_t
        _close("decode logits", torch.stack(rows[1:7]), G[p + "ref_decode_logits"], dt)
    else:
        n_same = next((i for i in range(6) if toks[i] != ref_t[i]), 6)
        assert n_same >= 1
        _close("decode logits", torch.stack(rows[1:1 + n_same]), G[p + "ref_decode_logits"][:n_same], dt, **tol)
    assert cache[0].offset == G[f"case{ci}.input_ids"].shape[1] + 7


def test_generate_step_text_and_image_prompts_vs_reference():
    """The reference's generate_step (ar.py:151-515), bf16 weights: text prompt bit for bit; image prompt with bf16 pixels
    (tokens equal, log-probs within the patch-embed summation-order noise) and with float32 pixels AS THE REFERENCE'S
    PIPELINE HANDS THEM OVER (the model never casts them: float32 tower + connector, rounded once by the scatter)."""
    cfg, W = oi.tiny_cfg(), _weights(torch.bfloat16)
    toks, rows = oi.generate_greedy(W, cfg, G["generate_step.text.input_ids"], None, max_tokens=6, return_logits=True)
    assert toks == G["generate_step.text.tokens"].tolist()
    lp = torch.stack([ops.logprobs_from_logits(r[None])[0] for r in rows]).to(torch.float32).numpy()
    assert np.array_equal(lp, G["generate_step.text.logprobs"])
    ids, pv, pm = G["case0.input_ids"], torch.from_numpy(G["case0.pixel_values"]), G["case0.pixel_attention_mask"]
    toks, rows = oi.generate_greedy(W, cfg, ids, pv, pm, max_tokens=6, return_logits=True)
    assert toks == G["generate_step.image.tokens"].tolist()
    lp = torch.stack([ops.logprobs_from_logits(r[None])[0] for r in rows]).to(torch.float32).numpy()
    ref = G["generate_step.image.logprobs"]
    assert np.abs(lp - ref).max() <= 4 * 2.0 ** -8 * np.abs(ref).max()
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    emb = oi.get_input_embeddings(W, cfg, ids, pv, pm, cast_pixels=False)
    assert emb.dtype == torch.bfloat16
    lg = oi.language_model(W, cfg, emb, cache, last_only=True)[0, -1]
    lp0 = ops.logprobs_from_logits(lg[None])[0].float().numpy()
    ref0 = G["generate_step.image_f32_pixels.logprobs"][0]
    assert int(lp0.argmax()) == int(G["generate_step.image_f32_pixels.tokens"][0])
    assert np.abs(lp0 - ref0).max() <= 4 * 2.0 ** -8 * np.abs(ref0).max()


def test_position_ids_bucketing_follows_the_reference_not_hf():
    """vision.py:143-166 as written: boundaries = linspace(1 / S, 1, S, endpoint=False) (S values 0.1, 0.19, ... - HF's are
    arange(1 / S, 1, 1 / S)) and bucket = digitize(x, boundaries, right=True) - 1, i.e. (number of boundaries below x) - 1:
    the first coordinates land in bucket -1, and a NEGATIVE position id indexes the table from its end (mx / python
    indexing).  The oracle restates that arithmetic (bit-exact tower outputs above); here the rule itself on small grids."""
    S = 10
    bounds = [1 / S + (1 - 1 / S) / S * j for j in range(S)]

    def bucket(x):
        return sum(b < x for b in bounds) - 1

    full = np.ones((1, 10, 10), dtype=bool)
    want = [[bucket(r / 10) * S + bucket(c / 10) for c in range(10)] for r in range(10)]
    assert oi.position_ids(full, S)[0].reshape(10, 10).tolist() == want and want[0][0] == -11 and want[9][9] == 88
    m = np.zeros((1, 6, 7), dtype=bool)
    m[0, :5, :5] = True
    ids = oi.position_ids(m, S)[0].reshape(6, 7)
    assert ids[:5, :5].tolist() == [[bucket(r / 5) * S + bucket(c / 5) for c in range(5)] for r in range(5)]
    assert not ids[5].any() and not ids[:, 5:].any()                    # dead patches keep id 0


def test_sanitize_key_map_vs_reference():
    from mlx_vlm_amd.models.idefics2 import sanitize_keys

    assert sorted(sanitize_keys(list(G["sanitize.keys_in"]))) == list(G["sanitize.keys_out"])
