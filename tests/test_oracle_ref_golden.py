"""Pin the oracle against vectors produced by the REFERENCE'S OWN Python files
(tests/golden/qwen2_vl_tiny_ref.npz, made by tests/golden/make_golden_ref.py, which executes
/root/reference/mlx_vlm/models/qwen2_vl/*.py, models/{base,cache,rope_utils,mlp,activations}.py and
sample_utils.py over oracle/mlx_shim - a torch-CPU stand-in for the uninstallable `mlx`).

fp32: the restatement and the reference agree to accumulation-order noise.
bf16: the oracle's "typed graph" (where it rounds to bf16) is BIT-EXACT against the reference's graph on the
language-model path (prefill + KV-cache decode, pure-MLX rope path) and on the vision tower from the patch
embeddings on; the patch-embed contraction itself agrees to 1 bf16 ulp (Conv3d vs GEMM summation order).
CPU only; nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ops
from oracle import qwen2_vl as oq

HERE = os.path.dirname(__file__)
R = np.load(os.path.join(HERE, "golden", "qwen2_vl_tiny_ref.npz"))
G = np.load(os.path.join(HERE, "golden", "qwen2_vl_tiny_hf.npz"))
CASES = ["one_image", "two_images"]
DT = {"f32": torch.float32, "bf16": torch.bfloat16}


def weights(dt):
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=torch.float32, std=0.05, embed_std=0.2)
    return cfg, {k: v.to(dt) for k, v in W.items()}


@pytest.mark.parametrize("case", CASES)
def test_reference_over_shim_agrees_with_huggingface_fp32(case):
    """Fidelity of the stand-in itself: the reference's code over oracle/mlx_shim vs HF transformers fp32."""
    np.testing.assert_allclose(R[case + ".f32.ref_image_features"], G[case + ".hf_image_features"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(R[case + ".f32.ref_prefill_logits"], G[case + ".hf_logits"], rtol=2e-4, atol=2e-4)
    assert R[case + ".f32.ref_greedy"].tolist() == G[case + ".hf_greedy"].tolist()
    assert np.array_equal(R[case + ".input_ids"], G[case + ".input_ids"])


@pytest.mark.parametrize("case", CASES)
def test_rope_index_matches_reference(case):
    cfg = oq.tiny_cfg()
    pos, delta = oq.get_rope_index(cfg, R[case + ".input_ids"], R[case + ".grid_thw"])
    assert np.array_equal(np.asarray(pos), R[case + ".ref_position_ids"])
    assert np.asarray(delta).reshape(-1).tolist() == R[case + ".ref_rope_deltas"].tolist()


def test_rope_index_text_and_padded_match_reference():
    cfg = oq.tiny_cfg()
    ids, am = R["text_padded.input_ids"], R["text_padded.attention_mask"]
    pos, delta = oq.get_rope_index(cfg, ids, attention_mask=am)
    assert np.array_equal(np.asarray(pos), R["text_padded.ref_position_ids"])
    assert np.asarray(delta).reshape(-1).tolist() == R["text_padded.ref_rope_deltas"].tolist()
    pos, delta = oq.get_rope_index(cfg, ids)
    assert np.array_equal(np.asarray(pos), R["text_only.ref_position_ids"])
    assert np.asarray(delta).reshape(-1).tolist() == R["text_only.ref_rope_deltas"].tolist()
    pos, delta = oq.get_rope_index(cfg, R["image_padded.input_ids"], R["image_padded.grid_thw"], None,
                                   R["image_padded.attention_mask"])
    assert np.array_equal(np.asarray(pos), R["image_padded.ref_position_ids"])
    assert np.asarray(delta).reshape(-1).tolist() == R["image_padded.ref_rope_deltas"].tolist()


@pytest.mark.parametrize("case", CASES)
def test_fp32_forward_matches_reference(case):
    cfg, W = weights(torch.float32)
    pix, thw, ids = torch.from_numpy(G[case + ".pixel_values"]), R[case + ".grid_thw"], R[case + ".input_ids"]
    feats = oq.vision_tower(W, cfg, pix, thw)
    np.testing.assert_allclose(feats.numpy(), R[case + ".f32.ref_image_features"], rtol=2e-5, atol=2e-5)
    emb, pos, _ = oq.get_input_embeddings(W, cfg, ids, pix, thw)
    np.testing.assert_allclose(emb[0].numpy(), R[case + ".f32.ref_inputs_embeds"], rtol=2e-5, atol=2e-5)
    toks, lg = oq.generate_greedy(W, cfg, ids, pix, thw, max_tokens=9, rope_mode="fallback", return_logits=True)
    assert toks[:8] == R[case + ".f32.ref_greedy"].tolist()
    np.testing.assert_allclose(lg[0].numpy(), R[case + ".f32.ref_prefill_logits"][-1], rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(lg[1:].numpy(), R[case + ".f32.ref_decode_logits"], rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("case", CASES)
def test_bf16_language_path_is_bit_exact_vs_reference_graph(case):
    """Same merged embeddings in -> every prefill logit and every KV-cache decode step bit-identical: pins the
    oracle's rounding points (RMSNorm, qkv+bias, pure-MLX M-RoPE, SDPA, SwiGLU, residuals, tied head), the
    KVCache and the decode position rule (cache offset + rope_delta, language.py:476-509)."""
    cfg, W = weights(torch.bfloat16)
    p = case + ".bf16."
    emb = torch.from_numpy(R[p + "ref_inputs_embeds"]).to(torch.bfloat16)[None]
    pos = torch.from_numpy(R[case + ".ref_position_ids"])
    delta = int(R[case + ".ref_rope_deltas"][0])
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    lg = oq.lm_head(W, cfg, oq.qwen2_model(W, cfg, emb, cache, pos, "fallback"))[0]
    assert np.array_equal(lg.float().numpy(), R[p + "ref_prefill_logits"])
    y = int(ops.argmax_first(ops.logprobs_from_logits(lg[-1:]))[0])
    for n in range(8):
        assert y == int(R[p + "ref_greedy"][n])
        e = oq.embed_tokens(W, np.array([[y]]))
        pid = torch.full((3, 1, 1), cache[0].offset + delta, dtype=torch.long)
        l = oq.lm_head(W, cfg, oq.qwen2_model(W, cfg, e, cache, pid, "fallback"))[0, -1]
        assert np.array_equal(l.float().numpy(), R[p + "ref_decode_logits"][n]), f"decode step {n}"
        y = int(ops.argmax_first(ops.logprobs_from_logits(l[None]))[0])
    assert cache[0].offset == int(R[p + "ref_kv_offset"][0])


@pytest.mark.parametrize("case", CASES)
def test_bf16_vision_tower_is_bit_exact_from_patch_embeddings(case):
    cfg, W = weights(torch.bfloat16)
    p = case + ".bf16."
    pix = torch.from_numpy(G[case + ".pixel_values"]).to(torch.bfloat16)
    x0 = oq.patch_embed(W, cfg, pix).float().numpy()
    ref0 = R[p + "ref_patch_embed"]
    # Conv3d vs GEMM: same fp32 products, different summation order -> at most one bf16 ulp (2^-8 relative), rarely
    assert np.all(np.abs(x0 - ref0) <= 2.0 ** -7 * np.abs(ref0) + 1e-6)
    assert (x0 != ref0).mean() < 0.01
    feats = oq.vision_tower(W, cfg, None, R[case + ".grid_thw"], patch_embeds=torch.from_numpy(ref0).to(torch.bfloat16))
    assert np.array_equal(feats.float().numpy(), R[p + "ref_image_features"])
    # the merge (masked gather + where, qwen2_vl.py:78-148) given the reference's features
    ids = R[case + ".input_ids"]
    emb = oq.merge_input_ids_with_image_features(cfg, torch.from_numpy(R[p + "ref_image_features"]).to(torch.bfloat16),
                                                 oq.embed_tokens(W, ids), ids)
    assert np.array_equal(emb[0].float().numpy(), R[p + "ref_inputs_embeds"])


@pytest.mark.parametrize("case", CASES)
def test_bf16_end_to_end_close_and_same_tokens(case):
    """End to end (image -> tokens) the only non-identical op is the patch-embed summation order."""
    cfg, W = weights(torch.bfloat16)
    p = case + ".bf16."
    pix, thw, ids = torch.from_numpy(G[case + ".pixel_values"]), R[case + ".grid_thw"], R[case + ".input_ids"]
    toks, lg = oq.generate_greedy(W, cfg, ids, pix, thw, max_tokens=9, rope_mode="fallback", return_logits=True)
    assert toks[:8] == R[p + "ref_greedy"].tolist()
    ref = np.concatenate([R[p + "ref_prefill_logits"][-1:], R[p + "ref_decode_logits"]])
    err = np.abs(lg.float().numpy() - ref)
    assert err.max() <= 4 * 2.0 ** -7 * np.abs(ref).max()


def test_sampler_filters_match_reference():
    lp = torch.from_numpy(R["sampler.logprobs"])
    assert np.array_equal(ops.apply_top_k(lp, 5).numpy(), R["sampler.top_k_5"])
    assert np.array_equal(ops.apply_top_p(lp, 0.9).numpy(), R["sampler.top_p_0.9"])
    assert np.array_equal(ops.apply_top_p(lp, 0.5).numpy(), R["sampler.top_p_0.5"])
    assert np.array_equal(ops.apply_min_p(lp, 0.05).numpy(), R["sampler.min_p_0.05"])
    assert ops.argmax_first(lp).tolist() == R["sampler.greedy"].tolist()


def test_fused_rope_mode_stays_within_reference_self_consistency():
    """The Metal kernel cannot run off-Metal; the reference's own contract for it is fused == pure-MLX within
    atol 1e-4 in fp32 (reference tests/test_rope_utils.py:366-407).  Same contract for the oracle's two modes."""
    cfg, W = weights(torch.float32)
    case = "one_image"
    emb = torch.from_numpy(R[case + ".f32.ref_inputs_embeds"])[None]
    pos = torch.from_numpy(R[case + ".ref_position_ids"])
    a = oq.qwen2_model(W, cfg, emb, None, pos, "fused")
    b = oq.qwen2_model(W, cfg, emb, None, pos, "fallback")
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=1e-4, rtol=0)


def test_generate_step_text_prompt_is_bit_exact_vs_reference():
    """The reference's generate_step (generate/ar.py:151-515) itself, greedy, bf16, text-only prompt: tokens and
    every bf16 logprob (logits - logsumexp, ar.py:368) identical."""
    cfg, W = weights(torch.bfloat16)
    ids = R["generate_step.text.input_ids"]
    toks, lg = oq.generate_greedy(W, cfg, ids, None, None, max_tokens=8, rope_mode="fallback", return_logits=True)
    assert toks == R["generate_step.text.tokens"].tolist()
    lp = ops.logprobs_from_logits(lg)
    assert lp.dtype == torch.bfloat16
    assert np.array_equal(lp.float().numpy(), R["generate_step.text.logprobs"])


def test_generate_step_image_prompt_matches_reference():
    cfg, W = weights(torch.bfloat16)
    case = "one_image"
    pix, thw, ids = torch.from_numpy(G[case + ".pixel_values"]), R[case + ".grid_thw"], R[case + ".input_ids"]
    toks, lg = oq.generate_greedy(W, cfg, ids, pix, thw, max_tokens=8, rope_mode="fallback", return_logits=True)
    assert toks == R["generate_step.image.tokens"].tolist()
    lp = ops.logprobs_from_logits(lg).float().numpy()
    ref = R["generate_step.image.logprobs"]
    assert np.abs(lp - ref).max() <= 4 * 2.0 ** -7 * np.abs(ref).max()      # patch-embed summation order only


@pytest.mark.parametrize("tag", ["ip_a", "ip_b"])
def test_image_processor_is_bit_exact_vs_reference(tag):
    """The reference's own Qwen3VLImageProcessor (processing_qwen3_vl.py:302-378, PIL bicubic) on the two test
    images: grid, every pixel row (crc32 of the float32 bytes) identical."""
    import zlib

    from oracle import image_processor as ip

    im = G[tag + ".image_hwc"]
    pv, thw = ip.process([im.transpose(2, 0, 1)])
    assert thw.tolist() == R[tag + ".ref_grid_thw"].tolist()
    pv = np.ascontiguousarray(pv.astype(np.float32))
    assert zlib.crc32(pv.tobytes()) == int(R[tag + ".ref_pixel_values_crc32"][0])
    if R[tag + ".ref_pixel_values"].shape[0]:
        assert np.array_equal(pv, R[tag + ".ref_pixel_values"])


def test_smart_resize_matches_reference_table():
    from oracle import image_processor as ip

    for h, w, rh, rw in R["smart_resize.ref_table"].tolist():
        assert ip.smart_resize(h, w) == (rh, rw)


def test_checkpoint_sanitize_matches_reference():
    """Product Model.sanitize + VisionModel.sanitize (host code, CPU) vs the reference's own on the HF Qwen2-VL key
    layout: same key names in the same order, same conv-weight layout (O,C,T,H,W) -> (O,T,H,W,C)."""
    from mlx_vlm_amd.models.qwen2_vl.qwen2_vl import Model
    from mlx_vlm_amd.models.qwen2_vl.vision import VisionModel

    hf_keys = [str(k) for k in R["sanitize.hf_keys"]]
    w = {k: (torch.from_numpy(R["sanitize.hf_conv"]) if k.endswith("patch_embed.proj.weight") else torch.zeros(3))
         for k in hf_keys}
    san = VisionModel.sanitize(None, Model.sanitize(None, w))
    assert list(san.keys()) == [str(k) for k in R["sanitize.ref_keys"]]
    assert np.array_equal(san["vision_tower.patch_embed.proj.weight"].numpy(), R["sanitize.ref_conv"])
    # and the oracle's own sanitize
    o = oq.sanitize(w)
    assert set(o.keys()) == set(san.keys())
    assert np.array_equal(o["vision_tower.patch_embed.proj.weight"].numpy(), R["sanitize.ref_conv"])


def test_streaming_detokenizer_matches_reference():
    """Product NaiveStreamingDetokenizer vs the reference's (tokenizer_utils.py:71-118) segment by segment on a
    byte-level toy tokenizer (split multi-byte characters, newline flushes)."""
    from mlx_vlm_amd.utils import NaiveStreamingDetokenizer

    class ByteTok:
        def decode(self, toks):
            return bytes(toks).decode("utf-8", errors="replace")

    det = NaiveStreamingDetokenizer(ByteTok())
    det.reset()
    segs = []
    for t in R["detok.tokens"].tolist():
        det.add_token(t)
        segs.append(det.last_segment)
    det.finalize()
    segs.append(det.last_segment)
    assert segs == [str(x) for x in R["detok.ref_segments"]]
    assert det.text == str(R["detok.ref_text"][0])


def test_logits_processors_match_reference_make_logits_processors():
    """oracle/ops.py::apply_logits_processors against the reference's own make_logits_processors (sample_utils.py:92-146)
    run over the shim (tests/golden/make_golden_penalties.py): logit_bias, repetition / presence / frequency penalties
    with windows full of repeats, two rows - bit-exact bf16."""
    P = np.load(os.path.join(os.path.dirname(__file__), "golden", "penalties_ref.npz"))
    for ci in range(int(P["n_cases"])):
        g = lambda k: P[f"case{ci}.{k}"]                                      # noqa: E731
        f = lambda k: None if np.isnan(g(k)) else float(g(k))                  # noqa: E731
        bias = dict(zip(g("bias_idx").tolist(), g("bias_val").tolist())) or None
        x = torch.from_numpy(g("logits_bf16_as_f32")).to(torch.bfloat16)
        y = ops.apply_logits_processors(x, g("tokens"), bias, f("rep"), int(g("rep_ctx")), f("pres"), int(g("pres_ctx")),
                                      f("freq"), int(g("freq_ctx")))
        assert torch.equal(y, torch.from_numpy(g("out_bf16_as_f32")).to(torch.bfloat16)), ci
