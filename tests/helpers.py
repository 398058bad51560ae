"""Shared test helpers: build the product model (HIP) and the oracle from the same seeded weights."""
from __future__ import annotations

import numpy as np
import torch

from oracle import image_processor as oip
from oracle import qwen2_vl as oq


def model_config_from_oracle(cfg: oq.Cfg):
    from mlx_vlm_amd.models.qwen2_vl import ModelConfig

    t, v = cfg.text, cfg.vision
    d = dict(
        model_type="qwen2_vl", hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
        intermediate_size=t.intermediate_size, num_attention_heads=t.num_attention_heads,
        num_key_value_heads=t.num_key_value_heads, rms_norm_eps=t.rms_norm_eps, vocab_size=t.vocab_size,
        rope_theta=t.rope_theta, rope_scaling={"type": "mrope", "mrope_section": list(t.mrope_section)},
        tie_word_embeddings=t.tie_word_embeddings, image_token_id=cfg.image_token_id,
        video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_token_id,
        vision_config=dict(model_type="qwen2_vl", depth=v.depth, embed_dim=v.embed_dim, hidden_size=v.hidden_size,
                           num_heads=v.num_heads, patch_size=v.patch_size, mlp_ratio=v.mlp_ratio,
                           in_channels=v.in_channels, spatial_merge_size=v.spatial_merge_size,
                           temporal_patch_size=v.temporal_patch_size),
    )
    return ModelConfig.from_dict(d)


def build_product_model(cfg: oq.Cfg, W, device="cuda", **kw):
    from mlx_vlm_amd.models.qwen2_vl import Model

    m = Model(model_config_from_oracle(cfg), device=device, **kw)
    m.load_weights(W)
    return m


def synth_request(cfg: oq.Cfg, sizes, n_text=12, seed=0, text_hi=1000):
    """-> (input_ids [1,L] int64, pixel_values f32 [N,1176], grid_thw [n,3])"""
    rng = np.random.default_rng(seed)
    imgs = [rng.integers(0, 256, (3, h, w), dtype=np.uint8) for (h, w) in sizes]
    pix, thw = oip.process(imgs)
    ids = []
    for _ in imgs:
        ids += [cfg.vision_start_token_id, cfg.image_token_id, cfg.vision_start_token_id + 1]
    ids += rng.integers(3, text_hi, n_text).tolist()
    ids = oip.expand_image_placeholders(ids, cfg.image_token_id, thw)
    return np.array([ids], dtype=np.int64), pix, thw


def bf16_close(a: torch.Tensor, b: torch.Tensor, ulps: float = 2.0, frac_exact: float = 0.0, atol_rms: float = 2e-3):
    """bf16 comparison: |a-b| <= ulps * 2^-7 * |b| + atol_rms * rms(b) element-wise (one bf16 ulp is between
    2^-8 and 2^-7 of the value, so 2^-7 |b| bounds it from above); returns (ok, report)."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    rms = float(b.pow(2).mean().sqrt()) + 1e-30
    tol = ulps * (2.0 ** -7) * b.abs() + atol_rms * rms
    err = (a - b).abs()
    bad = err > tol
    nbad = int(bad.sum())
    exact = float((a == b).float().mean())
    rep = f"max_err={float(err.max()):.4g} rms={rms:.4g} bad={nbad}/{a.numel()} exact={exact:.3f}"
    ok = nbad == 0 and exact >= frac_exact and bool(torch.isfinite(a).all())
    return ok, rep


def build_bunny_model(cfg, W, device="cuda", **kw):
    """nanoLLaVA product model from oracle/llava_bunny.py's config + weights (already under the sanitized names)."""
    from mlx_vlm_amd.models.llava_bunny import Model, ModelConfig

    t, v = cfg.text, cfg.vision
    mc = ModelConfig.from_dict(dict(
        model_type="llava_bunny", auto_map={}, hidden_size=t.hidden_size, mm_hidden_size=v.hidden_size,
        num_hidden_layers=t.num_hidden_layers, intermediate_size=t.intermediate_size,
        num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
        rms_norm_eps=t.rms_norm_eps, vocab_size=t.vocab_size, rope_theta=t.rope_theta,
        attention_bias=t.attention_bias, tie_word_embeddings=t.tie_word_embeddings,
        image_token_index=cfg.image_token_index,
        vision_config=dict(num_hidden_layers=v.num_hidden_layers, hidden_size=v.hidden_size,
                           intermediate_size=v.intermediate_size, num_attention_heads=v.num_attention_heads,
                           image_size=v.image_size, patch_size=v.patch_size, num_channels=v.num_channels,
                           layer_norm_eps=v.layer_norm_eps)))
    m = Model(mc, device=device, **kw)
    weights = m.language_model.sanitize(dict(W))
    m.load_weights(weights)
    return m


def phi3v_config_from_oracle(cfg, quantization=None):
    """ModelConfig of models/phi3_v from oracle/phi3_v.py's Cfg (HF layout: text parameters at the root)"""
    from mlx_vlm_amd.models.phi3_v import ModelConfig

    t, v = cfg.text, cfg.vision
    d = dict(model_type="phi3_v", vocab_size=t.vocab_size, num_hidden_layers=t.num_hidden_layers,
             intermediate_size=t.intermediate_size, num_attention_heads=t.num_attention_heads,
             num_key_value_heads=t.num_key_value_heads, rms_norm_eps=t.rms_norm_eps, hidden_size=t.hidden_size,
             rope_theta=t.rope_theta, max_position_embeddings=t.max_position_embeddings,
             original_max_position_embeddings=t.original_max_position_embeddings,
             vision_config=dict(num_hidden_layers=v.num_hidden_layers, hidden_size=v.hidden_size,
                                intermediate_size=v.intermediate_size, num_attention_heads=v.num_attention_heads,
                                image_size=v.image_size, patch_size=v.patch_size, layer_norm_eps=v.layer_norm_eps))
    if t.short_factor is not None:
        d["rope_scaling"] = {"type": "su", "short_factor": list(t.short_factor), "long_factor": list(t.long_factor)}
    if quantization:
        d["quantization"] = quantization
    return ModelConfig.from_dict(d)


def build_phi3v_model(cfg, W, device="cuda", **kw):
    """Phi-3.5-vision product model from oracle/phi3_v.py's config + (checkpoint-style) weights"""
    from mlx_vlm_amd.models.phi3_v import Model

    m = Model(phi3v_config_from_oracle(cfg), device=device, **kw)
    m.load_weights(W)
    return m


def idefics2_config_from_oracle(cfg):
    from mlx_vlm_amd.models.idefics2 import ModelConfig

    t, v, p = cfg.text, cfg.vision, cfg.perceiver
    return ModelConfig.from_dict(dict(
        model_type="idefics2", image_token_id=cfg.image_token_id, vocab_size=t.vocab_size,
        text_config=dict(model_type="mistral", hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                         num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, rms_norm_eps=t.rms_norm_eps, vocab_size=t.vocab_size,
                         rope_theta=t.rope_theta),
        vision_config=dict(model_type="idefics2", hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                           num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                           num_channels=v.num_channels, image_size=v.image_size, patch_size=v.patch_size,
                           layer_norm_eps=v.layer_norm_eps),
        perceiver_config=dict(model_type="idefics2", num_key_value_heads=p.num_key_value_heads, resampler_depth=p.resampler_depth,
                              resampler_head_dim=p.resampler_head_dim, resampler_n_heads=p.resampler_n_heads,
                              resampler_n_latents=p.resampler_n_latents)))


def build_idefics2_model(cfg, W, device="cuda", **kw):
    """Idefics2 product model from oracle/idefics2.py's config + weights (sanitized names)"""
    from mlx_vlm_amd.models.idefics2 import Model

    m = Model(idefics2_config_from_oracle(cfg), device=device, **kw)
    m.load_weights(W)
    return m


def peaked_full_depth(W, embed_key: str, head_key: str, gamma: float = 0.15, stride: int = 389, n_cycle=None,
                      embed_gain: float = 50.0, branch_gain: float = 0.02):
    """SURVEY section 8d's "peaked head" for FULL-DEPTH models (the tiny-model form is oracle.qwen2_vl.peak_head).  With
    N(0, 0.02^2) weights 28-32 layers deep the residual branches (0.36 rms per element and layer at hidden 3584) swamp an
    embedding of 0.02 rms, so the construction scales: the embedding table x embed_gain (rows of norm ~ sqrt(D)), every
    o_proj / down_proj x branch_gain (the layers still run in full; their sum stays ~20 % of the stream), the head keeps
    its N(0, 0.02^2) noise (logit rms ~ 1.2) and row succ(t) = (t + stride) mod n_cycle gains gamma * E[t] / |E[t]|
    (a logit of ~ gamma * sqrt(D) ~ 9 against a noise maximum of ~ 5.5 over 152 k rows).  Greedy decoding then walks the
    permutation: token identity with the oracle can be demanded with no tie rule.  -> new dict (same tensor objects where
    nothing changed); the oracle run in the test asserts the margin before anything is compared."""
    W = dict(W)
    dt = W[embed_key].dtype
    E = W[embed_key].float() * embed_gain
    W[embed_key] = E.to(dt)
    for k in list(W):
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            W[k] = (W[k].float() * branch_gain).to(dt)
    n = n_cycle or E.shape[0]
    head = W[head_key].float().clone()
    src = torch.arange(n)
    En = W[embed_key][:n].float()
    head[(src + stride) % n] += gamma * En / En.norm(dim=-1, keepdim=True).clamp_min(1e-6)
    W[head_key] = head.to(dt)
    return W
