"""The oracle's nanoLLaVA / Idefics2 / Phi-3.5-vision restatements against HuggingFace transformers fp32
(tests/golden/families_hf.npz, made by tests/golden/make_golden_hf_families.py) - the second, independent pin beside the
reference-over-shim goldens (test_oracle_ref_golden_*.py), as test_oracle_golden.py is for Qwen2-VL.

HF and the oracle both run fp32 here; the bar is 1e-4 absolute on logits of magnitude 1.5-8 and on tower states of
magnitude 15-18 (measured: <= 2.2e-5 - two fp32 summation orders).  Where the reference itself departs from HF
(quick-GELU in the SigLIP / Idefics2 towers, Idefics2's position ids and unmasked encoder), the generator drives HF's
modules the reference's way and lists each departure; nothing is absorbed in the tolerance.
"""
import os
import zlib

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "families_hf.npz"))
F32 = torch.float32
ATOL = 1e-4


def _image(seed, shape, tag):
    img = np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)
    assert zlib.crc32(img.tobytes()) == int(G[tag + ".image_crc32"][0])
    return img


@pytest.mark.parametrize("case", ["two", "one"])
def test_idefics2_tower_connector_and_logits_match_hf(case):
    """idefics2/vision.py:84-205, idefics2.py:36-177 + the Mistral decoder.  `two`: images of different sizes in one prompt
    (the smaller one's padding patches are attended, as the reference does)."""
    from oracle import idefics2 as oi

    cfg = oi.tiny_cfg()
    W = oi.random_weights(cfg, seed=4321, dtype=F32, **oi.TEST_WEIGHT_SCALES)
    rng = np.random.default_rng(91)
    imgs = [rng.integers(0, 256, (90, 60, 3), dtype=np.uint8), rng.integers(0, 256, (56, 70, 3), dtype=np.uint8)]
    which = G[f"idefics2.{case}.which"].tolist()
    pv, pm = oi.preprocess([[imgs[i] for i in which]], shortest_edge=56, longest_edge=140)
    pvt = torch.as_tensor(np.asarray(pv), dtype=F32)
    real, pmask = oi.real_images_and_patch_mask(pvt, pm, cfg.vision.patch_size)
    np.testing.assert_allclose(oi.vision_tower(W, cfg, real, pmask).numpy(), G[f"idefics2.{case}.hf_tower"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(oi.image_features(W, cfg, pvt, pm).numpy(), G[f"idefics2.{case}.hf_image_features"], rtol=0, atol=ATOL)
    ids = G[f"idefics2.{case}.input_ids"]
    logits = oi.language_model(W, cfg, oi.get_input_embeddings(W, cfg, ids, pvt, pm))[0].numpy()
    np.testing.assert_allclose(logits, G[f"idefics2.{case}.hf_logits"], rtol=0, atol=ATOL)
    assert (logits.argmax(-1) == G[f"idefics2.{case}.hf_logits"].argmax(-1)).all()


def test_nanollava_siglip_tower_and_qwen2_decode_match_hf():
    """llava_bunny/vision.py:27-201 (last encoder state) and language.py:15-145 through the KV cache, 6 greedy tokens."""
    from oracle import llava_bunny as ob

    cfg = ob.tiny_cfg()
    W = ob.random_weights(cfg, seed=1234, dtype=F32, **ob.TEST_WEIGHT_SCALES)
    pix = torch.from_numpy(ob.preprocess([_image(17, (300, 420, 3), "nanollava")]))
    np.testing.assert_allclose(ob.vision_tower(W, cfg, pix).numpy(), G["nanollava.hf_tower_last_state"], rtol=0, atol=ATOL)
    ids = G["nanollava.input_ids"]
    prompt = ob.language_model(W, cfg, ob.get_input_embeddings(W, cfg, ids, pix))[0, -8:].numpy()
    np.testing.assert_allclose(prompt, G["nanollava.hf_prompt_logits_last8"], rtol=0, atol=ATOL)
    toks, rows = ob.generate_greedy(W, cfg, ids, pix, max_tokens=6, return_logits=True)
    assert toks == G["nanollava.hf_greedy"].tolist()
    np.testing.assert_allclose(rows.numpy(), G["nanollava.hf_greedy_logits"], rtol=0, atol=ATOL)


def test_phi3v_clip_tower_matches_hf():
    """phi3_v/vision.py:28-176: hidden state -2, class row dropped, of the global view and both tiles"""
    from oracle import phi3_v as op

    cfg = op.tiny_cfg()
    W = op.random_weights(cfg, seed=2024, dtype=F32, **op.TEST_WEIGHT_SCALES)
    pix = torch.as_tensor(op.preprocess([_image(23, (200, 500, 3), "phi3v")])[0])
    hs = op.clip_features(W, cfg, pix.reshape(-1, *pix.shape[-3:]))
    np.testing.assert_allclose(hs[:, ::6, ::4].numpy(), G["phi3v.hf_clip_state_m2_sub"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("regime", ["short", "long"])
def test_phi3v_decoder_with_su_rope_matches_hf_longrope(regime):
    """phi3_v.py:17-197 over rope_utils.py:96-189 vs Phi3ForCausalLM(rope_type="longrope"): the short factors (every
    position below original_max) and the long ones (original_max 16 < the 466-token prompt), prefill + 5 cached steps."""
    from oracle import phi3_v as op

    cfg = op.tiny_cfg()
    cfg.text.original_max_position_embeddings, cfg.text.max_position_embeddings = G[f"phi3v.{regime}.rope"].tolist()
    W = op.random_weights(cfg, seed=2024, dtype=F32, **op.TEST_WEIGHT_SCALES)
    pix, sizes = op.preprocess([_image(23, (200, 500, 3), "phi3v")])[:2]
    pix = torch.as_tensor(pix)
    ids = G[f"phi3v.{regime}.input_ids"]
    prompt = op.language_model(W, cfg, op.get_input_embeddings(W, cfg, ids, pix, sizes))[0, -8:].numpy()
    np.testing.assert_allclose(prompt, G[f"phi3v.{regime}.hf_prompt_logits_last8"], rtol=0, atol=ATOL)
    toks, rows = op.generate_greedy(W, cfg, ids, pix, sizes, max_tokens=6, return_logits=True)
    assert toks == G[f"phi3v.{regime}.hf_greedy"].tolist()
    np.testing.assert_allclose(rows.numpy(), G[f"phi3v.{regime}.hf_greedy_logits"], rtol=0, atol=ATOL)
