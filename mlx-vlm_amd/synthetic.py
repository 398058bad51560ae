"""Synthetic checkpoints at real dims (there is no network for
real ones): Linear / Embedding ~ N(0, 0.02^2), norm weights 1, biases 0, bf16,
generated directly on the device with a seeded torch generator."""
from __future__ import annotations

from typing import Dict

import zlib

import torch

QWEN2_VL_2B = dict(
    model_type="qwen2_vl", hidden_size=1536, num_hidden_layers=28, intermediate_size=8960, num_attention_heads=12,
    num_key_value_heads=2, rms_norm_eps=1e-6, vocab_size=151936, rope_theta=1000000.0,
    rope_scaling={"type": "mrope", "mrope_section": [16, 24, 24]}, tie_word_embeddings=True,
    image_token_id=151655, video_token_id=151656, vision_start_token_id=151652,
    vision_config=dict(model_type="qwen2_vl", depth=32, embed_dim=1280, hidden_size=1536, num_heads=16, patch_size=14,
                       mlp_ratio=4, in_channels=3, spatial_merge_size=2, temporal_patch_size=2),
)

QWEN2_VL_7B = dict(
    QWEN2_VL_2B, hidden_size=3584, intermediate_size=18944, num_attention_heads=28, num_key_value_heads=4,
    vocab_size=152064, tie_word_embeddings=False,
    vision_config=dict(QWEN2_VL_2B["vision_config"], hidden_size=3584),
)


# nanoLLaVA (qnguyen3/nanoLLaVA: Qwen1.5-0.5B + SigLIP-so400m/14-384), public HF config values - BASELINE configs[0]
NANOLLAVA = dict(
    model_type="llava_bunny", auto_map={}, hidden_size=1024, mm_hidden_size=1152, num_hidden_layers=24,
    intermediate_size=2816, num_attention_heads=16, num_key_value_heads=16, rms_norm_eps=1e-6, vocab_size=151936,
    rope_theta=1000000.0, attention_bias=True, tie_word_embeddings=True, image_token_index=-200,
    vision_config=dict(model_type="siglip_vision_model", num_hidden_layers=27, hidden_size=1152, intermediate_size=4304,
                       num_attention_heads=16, image_size=384, patch_size=14),
)


# Phi-3.5-vision-instruct (microsoft/Phi-3.5-vision-instruct: Phi-3-mini decoder + CLIP ViT-L/14-336), public HF config values -
# BASELINE configs[4].  The checkpoint's 48 short / long RoPE factors are not reproduced here (no network): stand-ins of the
# same range and shape (ascending, [1, 1.3] and [1, 64])
PHI35_VISION = dict(
    model_type="phi3_v", hidden_size=3072, num_hidden_layers=32, intermediate_size=8192, num_attention_heads=32,
    num_key_value_heads=32, rms_norm_eps=1e-5, vocab_size=32064, rope_theta=10000.0, max_position_embeddings=131072,
    original_max_position_embeddings=4096, tie_word_embeddings=False,
    rope_scaling={"type": "su", "short_factor": [round(1.0 + 0.3 * i / 47, 4) for i in range(48)],
                  "long_factor": [round(1.0 + 63.0 * (i / 47) ** 2, 4) for i in range(48)]},
    vision_config=dict(num_hidden_layers=24, hidden_size=1024, intermediate_size=4096, num_attention_heads=16, image_size=336,
                       patch_size=14, layer_norm_eps=1e-5),
)


# Idefics2-8B (HuggingFaceM4/idefics2-8b: SigLIP-so400m/14-980 tower + perceiver resampler + Mistral-7B), public HF config values -
# BASELINE configs[3]
IDEFICS2_8B = dict(
    model_type="idefics2", image_token_id=32001, vocab_size=32003,
    text_config=dict(model_type="mistral", hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                     num_key_value_heads=8, rms_norm_eps=1e-5, vocab_size=32003, rope_theta=10000.0),
    vision_config=dict(model_type="idefics2", hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                       num_channels=3, image_size=980, patch_size=14, layer_norm_eps=1e-6),
    perceiver_config=dict(model_type="idefics2", num_key_value_heads=4, resampler_depth=3, resampler_head_dim=96,
                          resampler_n_heads=16, resampler_n_latents=64),
)


def idefics2_weight_shapes(cfg) -> Dict[str, tuple]:
    """name -> shape for an idefics2 ModelConfig, sanitized names (reference idefics2.py:294-321)"""
    t, v, p = cfg.text_config, cfg.vision_config, cfg.perceiver_config
    E, I, D, TI = v.hidden_size, v.intermediate_size, t.hidden_size, t.intermediate_size
    hd = D // t.num_attention_heads
    s: Dict[str, tuple] = {"vision_model.embeddings.patch_embedding.weight": (E, v.patch_size, v.patch_size, v.num_channels),
                           "vision_model.embeddings.patch_embedding.bias": (E,),
                           "vision_model.embeddings.position_embedding.weight": ((v.image_size // v.patch_size) ** 2, E),
                           "vision_model.post_layernorm.weight": (E,), "vision_model.post_layernorm.bias": (E,)}
    for i in range(v.num_hidden_layers):
        q = f"vision_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[q + f"self_attn.{n}.weight"], s[q + f"self_attn.{n}.bias"] = (E, E), (E,)
        s.update({q + "layer_norm1.weight": (E,), q + "layer_norm1.bias": (E,), q + "layer_norm2.weight": (E,),
                  q + "layer_norm2.bias": (E,), q + "mlp.fc1.weight": (I, E), q + "mlp.fc1.bias": (I,),
                  q + "mlp.fc2.weight": (E, I), q + "mlp.fc2.bias": (E,)})
    s.update({"connector.modality_projection.gate_proj.weight": (TI, E), "connector.modality_projection.up_proj.weight": (TI, E),
              "connector.modality_projection.down_proj.weight": (D, TI), "connector.perceiver_resampler.latents": (p.resampler_n_latents, D),
              "connector.perceiver_resampler.norm.weight": (D,)})
    ph = p.resampler_head_dim
    for i in range(p.resampler_depth):
        q = f"connector.perceiver_resampler.layers.{i}."
        s.update({q + "input_latents_norm.weight": (D,), q + "input_context_norm.weight": (D,), q + "post_attention_layernorm.weight": (D,),
                  q + "self_attn.q_proj.weight": (p.resampler_n_heads * ph, D), q + "self_attn.k_proj.weight": (p.num_key_value_heads * ph, D),
                  q + "self_attn.v_proj.weight": (p.num_key_value_heads * ph, D), q + "self_attn.o_proj.weight": (D, p.resampler_n_heads * ph),
                  q + "mlp.gate_proj.weight": (4 * D, D), q + "mlp.up_proj.weight": (4 * D, D), q + "mlp.down_proj.weight": (D, 4 * D)})
    s["language_model.embed_tokens.weight"] = (t.vocab_size, D)
    for i in range(t.num_hidden_layers):
        q = f"language_model.layers.{i}."
        s.update({q + "input_layernorm.weight": (D,), q + "post_attention_layernorm.weight": (D,),
                  q + "self_attn.q_proj.weight": (t.num_attention_heads * hd, D), q + "self_attn.k_proj.weight": (t.num_key_value_heads * hd, D),
                  q + "self_attn.v_proj.weight": (t.num_key_value_heads * hd, D), q + "self_attn.o_proj.weight": (D, t.num_attention_heads * hd),
                  q + "mlp.gate_proj.weight": (TI, D), q + "mlp.up_proj.weight": (TI, D), q + "mlp.down_proj.weight": (D, TI)})
    s["language_model.norm.weight"] = (D,)
    s["language_model.lm_head.weight"] = (t.vocab_size, D)
    return s


def phi3v_weight_shapes(cfg) -> Dict[str, tuple]:
    """name -> shape for a phi3_v ModelConfig under the reference's (= HF) names (phi3_v.py:136-177, vision.py:113-206)"""
    v = cfg.vision_config
    E, I, D = v.hidden_size, v.intermediate_size, cfg.hidden_size
    hd = D // cfg.num_attention_heads
    C = "model.vision_embed_tokens.img_processor.vision_model."
    s: Dict[str, tuple] = {C + "embeddings.class_embedding": (E,),
                           C + "embeddings.patch_embedding.weight": (E, v.patch_size, v.patch_size, v.num_channels),
                           C + "embeddings.position_embedding.weight": ((v.image_size // v.patch_size) ** 2 + 1, E),
                           C + "pre_layrnorm.weight": (E,), C + "pre_layrnorm.bias": (E,),
                           C + "post_layernorm.weight": (E,), C + "post_layernorm.bias": (E,)}
    for i in range(v.num_hidden_layers):
        p = f"{C}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"], s[p + f"self_attn.{n}.bias"] = (E, E), (E,)
        s.update({p + "layer_norm1.weight": (E,), p + "layer_norm1.bias": (E,), p + "layer_norm2.weight": (E,),
                  p + "layer_norm2.bias": (E,), p + "mlp.fc1.weight": (I, E), p + "mlp.fc1.bias": (I,),
                  p + "mlp.fc2.weight": (E, I), p + "mlp.fc2.bias": (E,)})
    T = "model.vision_embed_tokens."
    s.update({T + "glb_GN": (1, 1, 4 * E), T + "sub_GN": (1, 1, 1, 4 * E), T + "img_projection.0.weight": (D, 4 * E),
              T + "img_projection.0.bias": (D,), T + "img_projection.2.weight": (D, D), T + "img_projection.2.bias": (D,),
              "model.embed_tokens.weight": (cfg.vocab_size, D)})
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        s.update({p + "input_layernorm.weight": (D,), p + "post_attention_layernorm.weight": (D,),
                  p + "self_attn.qkv_proj.weight": ((cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * hd, D),
                  p + "self_attn.o_proj.weight": (D, cfg.num_attention_heads * hd),
                  p + "mlp.gate_up_proj.weight": (2 * cfg.intermediate_size, D), p + "mlp.down_proj.weight": (D, cfg.intermediate_size)})
    s["model.norm.weight"] = (D,)
    s["lm_head.weight"] = (cfg.vocab_size, D)
    return s


def bunny_weight_shapes(cfg) -> Dict[str, tuple]:
    """name -> shape for a llava_bunny ModelConfig, sanitized names (reference llava_bunny.py:180-222; the unused
    pooling head of the tower is left out)."""
    t, v = cfg.text_config, cfg.vision_config
    E, I, D = v.hidden_size, v.intermediate_size, t.hidden_size
    hd = D // t.num_attention_heads
    V = "vision_tower.vision_tower.vision_model."
    s: Dict[str, tuple] = {V + "embeddings.patch_embedding.weight": (E, v.patch_size, v.patch_size, v.num_channels),
                           V + "embeddings.patch_embedding.bias": (E,),
                           V + "embeddings.position_embedding.weight": ((v.image_size // v.patch_size) ** 2, E)}
    for i in range(v.num_hidden_layers):
        p = f"{V}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"], s[p + f"self_attn.{n}.bias"] = (E, E), (E,)
        s.update({p + "layer_norm1.weight": (E,), p + "layer_norm1.bias": (E,), p + "layer_norm2.weight": (E,),
                  p + "layer_norm2.bias": (E,), p + "mlp.fc1.weight": (I, E), p + "mlp.fc1.bias": (I,),
                  p + "mlp.fc2.weight": (E, I), p + "mlp.fc2.bias": (E,)})
    s.update({"mm_projector.linear_1.weight": (D, E), "mm_projector.linear_1.bias": (D,),
              "mm_projector.linear_2.weight": (D, D), "mm_projector.linear_2.bias": (D,),
              "language_model.model.embed_tokens.weight": (t.vocab_size, D)})
    for i in range(t.num_hidden_layers):
        p = f"language_model.model.layers.{i}."
        s.update({p + "input_layernorm.weight": (D,), p + "post_attention_layernorm.weight": (D,),
                  p + "self_attn.q_proj.weight": (t.num_attention_heads * hd, D), p + "self_attn.q_proj.bias": (t.num_attention_heads * hd,),
                  p + "self_attn.k_proj.weight": (t.num_key_value_heads * hd, D), p + "self_attn.k_proj.bias": (t.num_key_value_heads * hd,),
                  p + "self_attn.v_proj.weight": (t.num_key_value_heads * hd, D), p + "self_attn.v_proj.bias": (t.num_key_value_heads * hd,),
                  p + "self_attn.o_proj.weight": (D, t.num_attention_heads * hd),
                  p + "mlp.gate_proj.weight": (t.intermediate_size, D), p + "mlp.up_proj.weight": (t.intermediate_size, D),
                  p + "mlp.down_proj.weight": (D, t.intermediate_size)})
    s["language_model.model.norm.weight"] = (D,)
    return s


def weight_shapes(cfg) -> Dict[str, tuple]:
    """name -> shape for a qwen2_vl ModelConfig, sanitized names (reference qwen2_vl.py:179-190)."""
    if getattr(cfg, "model_type", None) == "llava_bunny":
        return bunny_weight_shapes(cfg)
    if getattr(cfg, "model_type", None) == "phi3_v":
        return phi3v_weight_shapes(cfg)
    if getattr(cfg, "model_type", None) == "idefics2":
        return idefics2_weight_shapes(cfg)
    t, v = cfg.text_config, cfg.vision_config
    E, D = v.embed_dim, t.hidden_size
    hd = D // t.num_attention_heads
    H = int(E * v.mlp_ratio)
    M = E * v.spatial_merge_size ** 2
    pd = v.in_channels * v.temporal_patch_size * v.patch_size * v.patch_size
    s: Dict[str, tuple] = {"vision_tower.patch_embed.proj.weight": (E, pd)}
    for i in range(v.depth):
        p = f"vision_tower.blocks.{i}."
        s.update({p + "norm1.weight": (E,), p + "norm1.bias": (E,), p + "norm2.weight": (E,), p + "norm2.bias": (E,),
                  p + "attn.qkv.weight": (3 * E, E), p + "attn.qkv.bias": (3 * E,), p + "attn.proj.weight": (E, E),
                  p + "attn.proj.bias": (E,), p + "mlp.fc1.weight": (H, E), p + "mlp.fc1.bias": (H,),
                  p + "mlp.fc2.weight": (E, H), p + "mlp.fc2.bias": (E,)})
    p = "vision_tower.merger."
    s.update({p + "ln_q.weight": (E,), p + "ln_q.bias": (E,), p + "mlp.0.weight": (M, M), p + "mlp.0.bias": (M,),
              p + "mlp.2.weight": (v.hidden_size, M), p + "mlp.2.bias": (v.hidden_size,)})
    s["language_model.model.embed_tokens.weight"] = (t.vocab_size, D)
    for i in range(t.num_hidden_layers):
        p = f"language_model.model.layers.{i}."
        s.update({p + "input_layernorm.weight": (D,), p + "post_attention_layernorm.weight": (D,),
                  p + "self_attn.q_proj.weight": (t.num_attention_heads * hd, D), p + "self_attn.q_proj.bias": (t.num_attention_heads * hd,),
                  p + "self_attn.k_proj.weight": (t.num_key_value_heads * hd, D), p + "self_attn.k_proj.bias": (t.num_key_value_heads * hd,),
                  p + "self_attn.v_proj.weight": (t.num_key_value_heads * hd, D), p + "self_attn.v_proj.bias": (t.num_key_value_heads * hd,),
                  p + "self_attn.o_proj.weight": (D, t.num_attention_heads * hd),
                  p + "mlp.gate_proj.weight": (t.intermediate_size, D), p + "mlp.up_proj.weight": (t.intermediate_size, D),
                  p + "mlp.down_proj.weight": (D, t.intermediate_size)})
    s["language_model.model.norm.weight"] = (D,)
    if not t.tie_word_embeddings:
        s["language_model.lm_head.weight"] = (t.vocab_size, D)
    return s


def random_weights(cfg, seed: int = 0, device="cuda", dtype=torch.bfloat16, std: float = 0.02, fill: bool = True):
    """fill=False allocates only (receiving side of the RCCL weight broadcast)."""
    g = torch.Generator(device=device).manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for name, shape in weight_shapes(cfg).items():
        if not fill:
            W[name] = torch.empty(shape, dtype=dtype, device=device)
        elif name.endswith(("norm1.weight", "norm2.weight", "ln_q.weight", "layernorm.weight", "model.norm.weight",
                            "layrnorm.weight", "_norm.weight", ".norm.weight")):
            W[name] = torch.ones(shape, dtype=dtype, device=device)
        elif name.endswith(".bias"):
            W[name] = torch.zeros(shape, dtype=dtype, device=device)
        else:
            W[name] = (torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)
    return W


def quantize_random_(W: Dict[str, torch.Tensor], prefix: str = "language_model.", seed: int = 1, std: float = 0.02,
                     skip: tuple = ()):
    """Synthetic MLX affine 4-bit checkpoint IN PLACE (no quantizer is part of the product - the reference's lives in
    convert.py, out of scope): every 2-D `<prefix>...weight` with in % 64 == 0 becomes random nibbles (uint32 words as
    int32 bit patterns, [out, in / 8]) + `scales` / `biases` [out, in / 64] such that the dequantized weights are
    ~uniform with standard deviation `std` (step = std * sqrt(12) / 15, bias = -7.5 steps)."""
    for name in [k for k in W if k.startswith(prefix) and not k.startswith(skip or ("\0",)) and k.endswith(".weight")
                 and W[k].dim() == 2 and W[k].shape[1] % 64 == 0]:
        N, K = W[name].shape
        dev = W[name].device
        g = torch.Generator(device=dev).manual_seed(seed + (zlib.crc32(name.encode()) & 0xFFFF))
        path = name[: -len(".weight")]
        W[name] = torch.randint(-2 ** 31, 2 ** 31 - 1, (N, K // 8), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        step = std * (12 ** 0.5) / 15
        sc = (step * (0.75 + 0.5 * torch.rand(N, K // 64, generator=g, device=dev))).to(torch.bfloat16)
        W[path + ".scales"] = sc
        W[path + ".biases"] = (sc.float() * -7.5).to(torch.bfloat16)
    return W
