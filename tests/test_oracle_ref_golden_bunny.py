"""nanoLLaVA (`llava_bunny`): the oracle against vectors produced by the REFERENCE'S OWN files
(tests/golden/make_golden_ref_bunny.py ran mlx_vlm/models/llava_bunny/*.py, models/base.py, models/cache.py and
generate/ar.py unmodified over oracle/mlx_shim; only the .npz is read here).  CPU only."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import llava_bunny as ob

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "llava_bunny_tiny_ref.npz"))
ROWS = slice(None, None, 7)
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def _weights(dt):
    W = ob.random_weights(ob.tiny_cfg(), seed=4321, dtype=torch.float32, **ob.TEST_WEIGHT_SCALES)
    return {k: v.to(dt) for k, v in W.items()}


def _pixels(i):
    return torch.from_numpy(ob.preprocess([G[f"img{i}.image_hwc"]]))


def _close(name, got, ref, dt):
    got = got.detach().to(torch.float32).numpy()
    if dt == "f32":
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5, err_msg=name)
    else:
        assert np.array_equal(got, ref), (name, int((got != ref).sum()), float(np.abs(got - ref).max()))


@pytest.mark.parametrize("i", [0, 1])
def test_image_processor_bit_exact_vs_reference(i):
    """ImageProcessor.preprocess (llava_bunny.py:24-57): resize to 384 x 384 bicubic, 1/255, (x - 0.5) / 0.5."""
    pv = np.ascontiguousarray(ob.preprocess([G[f"img{i}.image_hwc"]])[0])
    assert pv.shape == (3, 384, 384) and pv.dtype == np.float32
    np.testing.assert_allclose(pv.astype(np.float64).sum(axis=(0, 2)), G[f"img{i}.ref_pixel_rowsum"], rtol=0, atol=1e-9)
    assert zlib.crc32(pv.tobytes()) == int(G[f"img{i}.ref_pixel_crc32"][0])


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("i", [0, 1])
def test_vision_tower_projector_and_splice_vs_reference(i, dt):
    """SigLIP embeddings / encoder layers / last state, mlp2x_gelu projector, <image> splice.  bf16: the patch-embed
    contraction agrees to 1 bf16 ulp in < 0.1 % of the elements (Conv2d vs GEMM summation order); from the reference's
    embeddings on, everything is bit-exact."""
    cfg, W = ob.tiny_cfg(), _weights(DT[dt])
    pix = _pixels(i).to(DT[dt])
    p = f"case{i}.{dt}."
    start = None
    if dt == "bf16":
        own = ob.vision_embeddings(W, cfg, pix)[0].to(torch.float32).numpy()
        ref = G[p + "ref_embeddings_full"]
        # 1 ulp of the conv output (the sum with the position embedding can cancel to a small value)
        assert np.abs(own - ref).max() <= 2.0 ** -7 * np.abs(ref).max() and (own != ref).mean() < 1e-3
        start = torch.from_numpy(ref).to(torch.bfloat16)[None]      # continue from the reference's own embeddings
    last, states = ob.vision_tower(W, cfg, pix, return_layers=True, embeddings=start)
    _close("embeddings", states[0][0, ROWS], G[p + "ref_embeddings"], dt)
    _close("layer0", states[1][0, ROWS], G[p + "ref_layer0"], dt)
    _close("vision_last", last[0, ROWS], G[p + "ref_vision_last"], dt)
    feats = ob.mm_projector(W, last)
    _close("image_features", feats[0, ROWS], G[p + "ref_image_features"], dt)
    emb = ob.get_input_embeddings(W, cfg, G[f"case{i}.input_ids"], pix, vision_embeds=start)
    assert emb.shape[1] == int(G[p + "ref_inputs_embeds_len"][0]) == G[f"case{i}.input_ids"].shape[1] + 728
    _close("inputs_embeds", emb[0, ROWS], G[p + "ref_inputs_embeds"], dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("i", [0, 1])
def test_prefill_and_kvcache_decode_vs_reference(i, dt):
    """Qwen1.5 decoder: nn.RoPE at cache offsets, q/k/v bias, KVCache growth, lm_head on every row; 6 greedy steps."""
    cfg, W = ob.tiny_cfg(), _weights(DT[dt])
    pix = _pixels(i).to(DT[dt])
    p = f"case{i}.{dt}."
    from oracle import ops

    start = torch.from_numpy(G[p + "ref_embeddings_full"]).to(torch.bfloat16)[None] if dt == "bf16" else None
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    logits = ob.language_model(W, cfg, ob.get_input_embeddings(W, cfg, G[f"case{i}.input_ids"], pix, start), cache)
    _close("prefill rows", logits[0][::97], G[p + "ref_prefill_logits_rows"], dt)
    _close("prefill last", logits[0, -1], G[p + "ref_prefill_logits_last"], dt)
    toks, rows = ob.generate_greedy(W, cfg, G[f"case{i}.input_ids"], pix, max_tokens=7, return_logits=True,
                                    vision_embeds=start)
    assert toks[:6] == G[p + "ref_greedy"].tolist()
    _close("decode logits", rows[1:7], G[p + "ref_decode_logits"], dt)
    assert cache[0].offset == G[f"case{i}.input_ids"].shape[1] + 728


def test_generate_step_text_prompt_vs_reference():
    """The reference's generate_step (ar.py:151-515) on a text prompt, bf16: tokens and bf16 logprobs bit for bit."""
    from oracle import ops

    cfg, W = ob.tiny_cfg(), _weights(torch.bfloat16)
    toks, rows = ob.generate_greedy(W, cfg, G["generate_step.text.input_ids"], None, max_tokens=6, return_logits=True)
    assert toks == G["generate_step.text.tokens"].tolist()
    lp = torch.stack([ops.logprobs_from_logits(r[None])[0] for r in rows]).to(torch.float32).numpy()
    assert np.array_equal(lp, G["generate_step.text.logprobs"])


def test_generate_step_image_prompt_reference_as_shipped_promotes_to_float32():
    """generate_step with float32 pixel values and bf16 weights, exactly as the reference's pipeline calls it
    (utils.py:2090-2091 -> ar.py:394): llava_bunny never casts the pixels, so MLX type promotion makes every activation
    after the patch embedding float32 (oracle: cast_pixels=False).  Tokens identical; float32 log-probs to 2e-4 (the
    Conv2d / GEMM summation-order difference is no longer hidden by a bf16 rounding)."""
    from oracle import ops

    cfg, W = ob.tiny_cfg(), _weights(torch.bfloat16)
    toks, rows = ob.generate_greedy(W, cfg, G["case0.input_ids"], _pixels(0), max_tokens=6, return_logits=True,
                                    cast_pixels=False)
    assert rows.dtype == torch.float32
    assert toks == G["generate_step.image.tokens"].tolist()
    lp = torch.stack([ops.logprobs_from_logits(r[None])[0] for r in rows]).numpy()
    np.testing.assert_allclose(lp, G["generate_step.image.logprobs"], rtol=2e-4, atol=2e-4)
    # the bf16 typed graph (pixels cast first) is a different computation: its logits are bf16
    _, cast_rows = ob.generate_greedy(W, cfg, G["case0.input_ids"], _pixels(0), max_tokens=2, return_logits=True)
    assert cast_rows.dtype == torch.bfloat16


def test_sanitize_key_map_vs_reference():
    """Model.sanitize (llava_bunny.py:180-222) + LanguageModel.sanitize (language.py:163-174) + the conv layout
    (vision.py:243-266), on HF-layout key names."""
    from mlx_vlm_amd.models.llava_bunny import sanitize_keys

    out = sanitize_keys(list(G["sanitize.keys_in"]), tie_word_embeddings=True)
    assert sorted(out) == list(G["sanitize.keys_out"])
