"""Oracle restatement of mlx_vlm.models.qwen2_vl (TEST INFRASTRUCTURE).

Follows /root/reference/mlx_vlm/models/qwen2_vl/{vision,language,qwen2_vl}.py
function by function on torch-CPU tensors.  Weights live in a flat dict with the
reference's *sanitized* key names (qwen2_vl.py:179-190, vision.py:292-310):
`vision_tower.*`, `language_model.model.*`, `language_model.lm_head.weight`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import math
import numpy as np
import torch

from . import ops

F32 = torch.float32


# --------------------------------------------------------------------------
# config (config.py:12-86); text dims come from the checkpoint's config.json
# --------------------------------------------------------------------------
@dataclass
class VisionCfg:
    depth: int = 32
    embed_dim: int = 1280
    hidden_size: int = 1536
    num_heads: int = 16
    patch_size: int = 14
    mlp_ratio: float = 4.0
    in_channels: int = 3
    spatial_merge_size: int = 2
    temporal_patch_size: int = 2


@dataclass
class TextCfg:
    hidden_size: int = 1536
    num_hidden_layers: int = 28
    intermediate_size: int = 8960
    num_attention_heads: int = 12
    num_key_value_heads: int = 2
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: List[int] = field(default_factory=lambda: [16, 24, 24])
    tie_word_embeddings: bool = True


@dataclass
class Cfg:
    text: TextCfg = field(default_factory=TextCfg)
    vision: VisionCfg = field(default_factory=VisionCfg)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652


def tiny_cfg(**over) -> Cfg:
    """Toy dims in the spirit of the reference's test_qwen2_vl
    (tests/test_models.py:4308-4373) but with head_dim multiples the HIP
    kernels support (vision head_dim 80, text head_dim 128)."""
    c = Cfg(
        text=TextCfg(hidden_size=256, num_hidden_layers=2, intermediate_size=512,
                     num_attention_heads=2, num_key_value_heads=1, vocab_size=1024,
                     mrope_section=[16, 24, 24]),
        vision=VisionCfg(depth=2, embed_dim=160, hidden_size=256, num_heads=2),
        image_token_id=1001, video_token_id=1002, vision_start_token_id=1003,
    )
    for k, v in over.items():
        setattr(c, k, v)
    return c


def random_weights(cfg: Cfg, seed: int = 0, dtype=torch.bfloat16, std: float = 0.02,
                   embed_std: Optional[float] = None, fast: bool = False) -> Dict[str, torch.Tensor]:
    """Synthetic checkpoint: Linear/Embedding ~ N(0, std^2),
    norm weights 1 (+small noise so they are exercised), biases small noise."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}

    n_fast = [0]

    def rn(*shape, s=std):
        if fast and math.prod(shape) >= (1 << 20):      # big matrices: threaded Philox streams (ops.fast_normal)
            n_fast[0] += 1
            return ops.fast_normal(shape, (seed, n_fast[0]), s, dtype)
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    v, t = cfg.vision, cfg.text
    E = v.embed_dim
    pd = v.in_channels * v.temporal_patch_size * v.patch_size * v.patch_size
    # channels-last conv weight as the reference holds it after sanitize
    W["vision_tower.patch_embed.proj.weight"] = rn(E, v.temporal_patch_size, v.patch_size, v.patch_size, v.in_channels)
    for i in range(v.depth):
        p = f"vision_tower.blocks.{i}."
        W[p + "norm1.weight"] = (1 + rn(E, s=0.05).float()).to(dtype)
        W[p + "norm1.bias"] = rn(E)
        W[p + "norm2.weight"] = (1 + rn(E, s=0.05).float()).to(dtype)
        W[p + "norm2.bias"] = rn(E)
        W[p + "attn.qkv.weight"] = rn(3 * E, E)
        W[p + "attn.qkv.bias"] = rn(3 * E)
        W[p + "attn.proj.weight"] = rn(E, E)
        W[p + "attn.proj.bias"] = rn(E)
        H = int(E * v.mlp_ratio)
        W[p + "mlp.fc1.weight"] = rn(H, E)
        W[p + "mlp.fc1.bias"] = rn(H)
        W[p + "mlp.fc2.weight"] = rn(E, H)
        W[p + "mlp.fc2.bias"] = rn(E)
    M = E * v.spatial_merge_size ** 2
    W["vision_tower.merger.ln_q.weight"] = (1 + rn(E, s=0.05).float()).to(dtype)
    W["vision_tower.merger.ln_q.bias"] = rn(E)
    W["vision_tower.merger.mlp.0.weight"] = rn(M, M)
    W["vision_tower.merger.mlp.0.bias"] = rn(M)
    W["vision_tower.merger.mlp.2.weight"] = rn(v.hidden_size, M)
    W["vision_tower.merger.mlp.2.bias"] = rn(v.hidden_size)

    D = t.hidden_size
    hd = D // t.num_attention_heads
    W["language_model.model.embed_tokens.weight"] = rn(t.vocab_size, D, s=embed_std or std)
    for i in range(t.num_hidden_layers):
        p = f"language_model.model.layers.{i}."
        W[p + "input_layernorm.weight"] = (1 + rn(D, s=0.05).float()).to(dtype)
        W[p + "post_attention_layernorm.weight"] = (1 + rn(D, s=0.05).float()).to(dtype)
        W[p + "self_attn.q_proj.weight"] = rn(t.num_attention_heads * hd, D)
        W[p + "self_attn.q_proj.bias"] = rn(t.num_attention_heads * hd)
        W[p + "self_attn.k_proj.weight"] = rn(t.num_key_value_heads * hd, D)
        W[p + "self_attn.k_proj.bias"] = rn(t.num_key_value_heads * hd)
        W[p + "self_attn.v_proj.weight"] = rn(t.num_key_value_heads * hd, D)
        W[p + "self_attn.v_proj.bias"] = rn(t.num_key_value_heads * hd)
        W[p + "self_attn.o_proj.weight"] = rn(D, t.num_attention_heads * hd)
        W[p + "mlp.gate_proj.weight"] = rn(t.intermediate_size, D)
        W[p + "mlp.up_proj.weight"] = rn(t.intermediate_size, D)
        W[p + "mlp.down_proj.weight"] = rn(D, t.intermediate_size)
    W["language_model.model.norm.weight"] = (1 + rn(D, s=0.05).float()).to(dtype)
    if not t.tie_word_embeddings:
        W["language_model.lm_head.weight"] = rn(t.vocab_size, D, s=embed_std or std)
    return W


def sanitize(hf_weights: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Model.sanitize (qwen2_vl.py:179-190) + VisionModel.sanitize
    (vision.py:292-310): visual -> vision_tower, model -> language_model.model,
    conv weight (O,C,T,H,W) -> (O,T,H,W,C).  Also accepts the transformers>=4.5x
    layout (`model.visual.*`, `model.language_model.*`)."""
    out = {}
    for k, v in hf_weights.items():
        if k.startswith("model.visual."):
            k = "visual." + k[len("model.visual."):]
        elif k.startswith("model.language_model."):
            k = "model." + k[len("model.language_model."):]
        if "vision_tower" not in k:
            k = k.replace("visual", "vision_tower")
        if "language_model" not in k:
            if "model" in k:
                k = k.replace("model", "language_model.model")
            elif "lm_head" in k:
                k = k.replace("lm_head", "language_model.lm_head")
        if "position_ids" in k:
            continue
        if "patch_embed.proj.weight" in k and v.dim() == 5 and v.shape[-1] != 3:
            v = v.permute(0, 2, 3, 4, 1).contiguous()
        out[k] = v
    return out


# --------------------------------------------------------------------------
# vision tower (vision.py)
# --------------------------------------------------------------------------
def patch_embed(W, cfg: Cfg, pixel_values):
    """PatchEmbed (vision.py:68-102): rows [N, C*T*ph*pw] (C-major) -> channels
    last -> Conv3d(k=s) == GEMM against the (O, T, H, W, C) weight."""
    v = cfg.vision
    x = pixel_values.reshape(-1, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size)
    x = x.permute(0, 2, 3, 4, 1).reshape(x.shape[0], -1)  # moveaxis(1, 4)
    w = W["vision_tower.patch_embed.proj.weight"].reshape(v.embed_dim, -1)
    return ops.linear(x, w)


def vision_cu_seqlens(grid_thw: np.ndarray) -> np.ndarray:
    """VisionModel.__call__ cu_seqlens (vision.py:266-279)."""
    lens = []
    for t, h, w in np.asarray(grid_thw).tolist():
        lens += [h * w] * t
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


def vision_attention(W, p, cfg: Cfg, x, cu_seqlens, freqs):
    """Attention (vision.py:123-161)."""
    H = cfg.vision.num_heads
    N = x.shape[0]
    qkv = ops.linear(x, W[p + "qkv.weight"], W[p + "qkv.bias"]).reshape(N, 3, H, -1)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]  # [N, H, D]
    q = ops.apply_rotary_pos_emb_vision(q, freqs)
    k = ops.apply_rotary_pos_emb_vision(k, freqs)
    D = q.shape[-1]
    outs = []
    for a, b in zip(cu_seqlens[:-1], cu_seqlens[1:]):
        qs = q[a:b].permute(1, 0, 2)[None]
        ks = k[a:b].permute(1, 0, 2)[None]
        vs = v[a:b].permute(1, 0, 2)[None]
        o = ops.sdpa(qs, ks, vs, scale=D ** -0.5)
        outs.append(o[0].permute(1, 0, 2))
    o = torch.cat(outs, dim=0).reshape(N, -1)
    return ops.linear(o, W[p + "proj.weight"], W[p + "proj.bias"])


def vision_block(W, i, cfg: Cfg, x, cu_seqlens, freqs):
    """Qwen2VLVisionBlock (vision.py:177-194)."""
    p = f"vision_tower.blocks.{i}."
    h = ops.layer_norm(x, W[p + "norm1.weight"], W[p + "norm1.bias"])
    x = ops.add(x, vision_attention(W, p + "attn.", cfg, h, cu_seqlens, freqs))
    h = ops.layer_norm(x, W[p + "norm2.weight"], W[p + "norm2.bias"])
    h = ops.gelu_fast(ops.linear(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"]))
    h = ops.linear(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])
    return ops.add(x, h)


def patch_merger(W, cfg: Cfg, x):
    """PatchMerger (vision.py:105-120) == the multimodal projector."""
    p = "vision_tower.merger."
    m = cfg.vision.embed_dim * cfg.vision.spatial_merge_size ** 2
    x = ops.layer_norm(x, W[p + "ln_q.weight"], W[p + "ln_q.bias"]).reshape(-1, m)
    x = ops.gelu_erf(ops.linear(x, W[p + "mlp.0.weight"], W[p + "mlp.0.bias"]))
    return ops.linear(x, W[p + "mlp.2.weight"], W[p + "mlp.2.bias"])


def vision_tower(W, cfg: Cfg, pixel_values, grid_thw, return_blocks: bool = False, patch_embeds=None):
    """VisionModel.__call__ (vision.py:257-290).  `patch_embeds` (tests only): start from given PatchEmbed outputs."""
    x = patch_embed(W, cfg, pixel_values) if patch_embeds is None else patch_embeds
    hd = cfg.vision.embed_dim // cfg.vision.num_heads
    freqs = ops.vision_rotary_freqs(grid_thw, hd, cfg.vision.spatial_merge_size)
    cu = vision_cu_seqlens(grid_thw)
    blocks = [x]
    for i in range(cfg.vision.depth):
        x = vision_block(W, i, cfg, x, cu, freqs)
        blocks.append(x)
    out = patch_merger(W, cfg, x)
    return (out, blocks) if return_blocks else out


# --------------------------------------------------------------------------
# language model (language.py)
# --------------------------------------------------------------------------
def get_rope_index(cfg: Cfg, input_ids: np.ndarray, image_grid_thw=None, video_grid_thw=None,
                   attention_mask: Optional[np.ndarray] = None):
    """LanguageModel.get_rope_index (language.py:216-402), integer-exact.
    -> position_ids int64 [3,B,L] (or [B,L] for the text-only branch),
       rope_deltas int64 [B,1]."""
    input_ids = np.asarray(input_ids)
    B, L = input_ids.shape
    ms = cfg.vision.spatial_merge_size
    if image_grid_thw is not None or video_grid_thw is not None:
        if attention_mask is None:
            attention_mask = np.ones_like(input_ids)
        position_ids = np.ones((3, B, L), dtype=np.int64)
        deltas = []
        image_index = video_index = 0
        for i in range(B):
            row_mask = attention_mask[i].tolist()
            toks = [t for t, keep in zip(input_ids[i].tolist(), row_mask) if keep == 1]
            vision_tokens = [toks[idx + 1] for idx, t in enumerate(toks[:-1]) if t == cfg.vision_start_token_id]
            image_nums = sum(t == cfg.image_token_id for t in vision_tokens)
            video_nums = sum(t == cfg.video_token_id for t in vision_tokens)
            pos_list: List[np.ndarray] = []
            st = 0
            remain_images, remain_videos = image_nums, video_nums
            for _ in range(image_nums + video_nums):
                if cfg.image_token_id in toks and remain_images > 0:
                    ed_image = toks.index(cfg.image_token_id, st)
                else:
                    ed_image = len(toks) + 1
                if cfg.video_token_id in toks and remain_videos > 0:
                    ed_video = toks.index(cfg.video_token_id, st)
                else:
                    ed_video = len(toks) + 1
                if ed_image < ed_video:
                    t, h, w = [int(x) for x in image_grid_thw[image_index]]
                    image_index += 1
                    remain_images -= 1
                    ed = ed_image
                else:
                    t, h, w = [int(x) for x in video_grid_thw[video_index]]
                    video_index += 1
                    remain_videos -= 1
                    ed = ed_video
                gt, gh, gw = t, h // ms, w // ms
                text_len = ed - st
                st_idx = int(pos_list[-1].max()) + 1 if pos_list else 0
                pos_list.append(np.broadcast_to(np.arange(text_len)[None], (3, text_len)) + st_idx)
                ti = np.broadcast_to(np.arange(gt)[:, None], (gt, gh * gw)).reshape(-1)
                hi = np.broadcast_to(np.arange(gh)[None, :, None], (gt, gh, gw)).reshape(-1)
                wi = np.broadcast_to(np.arange(gw)[None, None, :], (gt, gh, gw)).reshape(-1)
                pos_list.append(np.stack([ti, hi, wi]) + text_len + st_idx)
                st = ed + gt * gh * gw
            if st < len(toks):
                st_idx = int(pos_list[-1].max()) + 1 if pos_list else 0
                text_len = len(toks) - st
                pos_list.append(np.broadcast_to(np.arange(text_len)[None], (3, text_len)) + st_idx)
            if not pos_list:
                deltas.append(0)
                continue
            llm_positions = np.concatenate(pos_list, axis=1).reshape(3, -1)
            compact_max = int(llm_positions.max())
            padded = np.ones((3, L), dtype=np.int64)
            keep_cols = [c for c, keep in enumerate(row_mask) if keep == 1]
            padded[:, keep_cols] = llm_positions
            position_ids[:, i, :] = padded
            deltas.append(compact_max + 1 - len(toks))
        return position_ids, np.array(deltas, dtype=np.int64).reshape(-1, 1)
    if attention_mask is not None:
        am = np.asarray(attention_mask).astype(np.int64)
        position_ids = np.cumsum(am, axis=-1) - 1
        position_ids = np.where(am == 0, 1, position_ids)
        deltas = position_ids.max(axis=-1, keepdims=True) + 1 - am.shape[-1]
        return position_ids, deltas
    position_ids = np.broadcast_to(np.arange(L)[None], (B, L)).astype(np.int64)
    return position_ids, np.zeros((B, 1), dtype=np.int64)


def llm_attention(W, p, cfg: Cfg, x, cache: Optional[ops.KVCache], position_ids, causal: bool,
                  rope_mode: str = "fused"):
    """Attention (language.py:40-120)."""
    t = cfg.text
    B, L, D = x.shape
    H, Hkv = t.num_attention_heads, t.num_key_value_heads
    hd = D // H
    q = ops.linear(x, W[p + "q_proj.weight"], W[p + "q_proj.bias"]).reshape(B, L, H, hd).permute(0, 2, 1, 3)
    k = ops.linear(x, W[p + "k_proj.weight"], W[p + "k_proj.bias"]).reshape(B, L, Hkv, hd).permute(0, 2, 1, 3)
    v = ops.linear(x, W[p + "v_proj.weight"], W[p + "v_proj.bias"]).reshape(B, L, Hkv, hd).permute(0, 2, 1, 3)
    if position_ids is None:
        off = cache.offset if cache is not None else 0
        position_ids = torch.arange(off, off + L)[None, None].expand(3, B, L)
    inv = ops.mrope_inv_freq(hd, t.rope_theta)
    sel = ops.chunked_position_selector(t.mrope_section, hd // 2)
    q = ops.mrope_apply(q, position_ids, inv, sel, rope_mode)
    k = ops.mrope_apply(k, position_ids, inv, sel, rope_mode)
    if cache is not None:
        k, v = cache.update_and_fetch(k, v)
    if hasattr(cache, "bits"):
        # base.py:356-365: a cache with `bits` (QuantizedKVCache) routes to quantized_scaled_dot_product_attention
        from . import quant
        o = quant.quantized_sdpa(q, k, v, scale=hd ** -0.5, causal=causal and L > 1, group_size=cache.group_size, bits=cache.bits)
    else:
        o = ops.sdpa(q, k, v, scale=hd ** -0.5, causal=causal, q_offset=k.shape[2] - L)
    o = o.permute(0, 2, 1, 3).reshape(B, L, -1)
    return ops.linear(o, W[p + "o_proj.weight"])


def decoder_layer(W, i, cfg: Cfg, x, cache, position_ids, causal, rope_mode="fused"):
    """Qwen2VLDecoderLayer (language.py:123-154)."""
    p = f"language_model.model.layers.{i}."
    r = llm_attention(W, p + "self_attn.", cfg, ops.rms_norm(x, W[p + "input_layernorm.weight"], cfg.text.rms_norm_eps),
                      cache, position_ids, causal, rope_mode)
    h = ops.add(x, r)
    hn = ops.rms_norm(h, W[p + "post_attention_layernorm.weight"], cfg.text.rms_norm_eps)
    g = ops.linear(hn, W[p + "mlp.gate_proj.weight"])
    u = ops.linear(hn, W[p + "mlp.up_proj.weight"])
    r = ops.linear(ops.swiglu(g, u), W[p + "mlp.down_proj.weight"])
    return ops.add(h, r)


def qwen2_model(W, cfg: Cfg, inputs_embeds, cache, position_ids, rope_mode="fused", return_layers=False):
    """Qwen2Model (language.py:157-200): mask "causal" when L > 1 else None
    (base.py:214-228)."""
    h = inputs_embeds
    L = h.shape[1]
    if cache is None:
        cache = [None] * cfg.text.num_hidden_layers
    layers = []
    for i in range(cfg.text.num_hidden_layers):
        h = decoder_layer(W, i, cfg, h, cache[i], position_ids, causal=L > 1, rope_mode=rope_mode)
        layers.append(h)
    out = ops.rms_norm(h, W["language_model.model.norm.weight"], cfg.text.rms_norm_eps)
    return (out, layers) if return_layers else out


def lm_head(W, cfg: Cfg, h):
    """embed_tokens.as_linear / lm_head (language.py:514-517)."""
    w = W["language_model.model.embed_tokens.weight"] if cfg.text.tie_word_embeddings else W["language_model.lm_head.weight"]
    return ops.linear(h, w)


def embed_tokens(W, input_ids):
    """nn.Embedding (language.py:164,179); nn.QuantizedEmbedding for a 4-bit checkpoint (dequantize of the rows)."""
    e = W["language_model.model.embed_tokens.weight"]
    idx = torch.as_tensor(np.asarray(input_ids), dtype=torch.long)
    return e.rows(idx) if hasattr(e, "wq") else e[idx]


def merge_input_ids_with_image_features(cfg: Cfg, image_features, inputs_embeds, input_ids):
    """Model.merge_input_ids_with_image_features (qwen2_vl.py:78-148)."""
    ids = np.asarray(input_ids)
    pos = ids == cfg.image_token_id
    if pos.sum() == 0:
        pos = ids == cfg.video_token_id
    out = inputs_embeds.clone()
    start = 0
    for b in range(ids.shape[0]):
        n = int(pos[b].sum())
        if n > 0:
            feats = image_features[start:start + n]
            if feats.shape[0] != n:
                raise ValueError(
                    f"Number of image token positions ({n}) does not match number of image features ({feats.shape[0]}) for batch {b}")
            out[b, torch.from_numpy(pos[b])] = feats.to(out.dtype)
            start += n
    return out


def get_input_embeddings(W, cfg: Cfg, input_ids, pixel_values=None, image_grid_thw=None, mask=None):
    """Model.get_input_embeddings (qwen2_vl.py:20-76).
    -> inputs_embeds [B,L,D], position_ids, rope_deltas."""
    emb = embed_tokens(W, input_ids)
    if pixel_values is None:
        pos, deltas = get_rope_index(cfg, input_ids, attention_mask=mask)
        return emb, pos, deltas
    dtype = W["vision_tower.patch_embed.proj.weight"].dtype
    feats = vision_tower(W, cfg, pixel_values.to(dtype), image_grid_thw)
    emb = merge_input_ids_with_image_features(cfg, feats, emb, input_ids)
    pos, deltas = get_rope_index(cfg, input_ids, image_grid_thw, None, mask)
    return emb, pos, deltas


def _make_prompt_cache(cfg: Cfg, max_kv_size: Optional[int] = None):
    """make_prompt_cache (cache.py:45-70): the built families define no make_cache, so `max_kv_size` gives every layer a
    RotatingKVCache(max_size, keep=4)"""
    n = cfg.text.num_hidden_layers
    return [ops.RotatingKVCache(max_kv_size, keep=4) for _ in range(n)] if max_kv_size is not None else [ops.KVCache() for _ in range(n)]


def _cache_offset(c0) -> int:
    """what LanguageModel.__call__ reads as the cache offset (language.py:426-431): the WRITE INDEX of a rotating cache"""
    return c0._idx if hasattr(c0, "_idx") else c0.offset


# --------------------------------------------------------------------------
# generate_step, greedy (generate/ar.py:151-515)
# --------------------------------------------------------------------------
def generate_greedy(W, cfg: Cfg, input_ids, pixel_values=None, image_grid_thw=None,
                    max_tokens: int = 16, rope_mode: str = "fused", return_logits: bool = False, processors=None,
                    kv_bits=None, kv_group_size: int = 64, quantized_kv_start: int = 0, return_logprobs: bool = False,
                    max_kv_size: Optional[int] = None):
    """generate_step with temperature 0: embeds -> full-prompt prefill ->
    logits[:, -1] -> logprobs = logits - logsumexp -> argmax -> decode loop with
    pos = cache offset + rope_delta (language.py:476-509)."""
    input_ids = np.asarray(input_ids)
    assert input_ids.shape[0] == 1
    emb, pos, deltas = get_input_embeddings(W, cfg, input_ids, pixel_values, image_grid_thw)
    pos_t = torch.from_numpy(np.asarray(pos))
    cache = _make_prompt_cache(cfg, max_kv_size)
    from . import quant
    h = qwen2_model(W, cfg, emb, cache, pos_t, rope_mode)
    logits = lm_head(W, cfg, h)[:, -1, :]
    quant.maybe_quantize_kv_cache(cache, quantized_kv_start, kv_group_size, kv_bits)    # ar.py:362: inside _step, after the forward
    toks, all_logits, all_lp = [], [], []
    delta = int(deltas[0, 0])
    fed = list(input_ids.reshape(-1))           # ar.py:360-364: `tokens` = the prompt, then every token fed back
    for n in range(max_tokens):
        if processors:
            logits = ops.apply_logits_processors(logits, fed, **processors)
        lp = ops.logprobs_from_logits(logits)
        y = int(ops.argmax_first(lp)[0])
        toks.append(y)
        all_logits.append(logits[0].clone())
        all_lp.append(lp[0].clone())
        if n == max_tokens - 1:
            break
        e = embed_tokens(W, np.array([[y]]))
        fed.append(y)
        p = _cache_offset(cache[0]) + delta
        pid = torch.full((3, 1, 1), p, dtype=torch.long)
        h = qwen2_model(W, cfg, e, cache, pid, rope_mode)
        logits = lm_head(W, cfg, h)[:, -1, :]
        quant.maybe_quantize_kv_cache(cache, quantized_kv_start, kv_group_size, kv_bits)
    if return_logprobs:
        return toks, torch.stack(all_lp)
    if return_logits:
        return toks, torch.stack(all_logits)
    return toks


# --------------------------------------------------------------------------
# test constructions on top of the restatement (no reference counterpart)
# --------------------------------------------------------------------------
def peak_head(W, cfg: Cfg, gamma: float = 1.0, stride: int = 389, n_cycle: Optional[int] = None):
    """SURVEY.md par. 8d "peaked head": an UNTIED lm_head whose row succ(t) = (t + stride) mod n_cycle carries
    gamma * E[t] / |E[t]| on top of the seeded noise.  The residual stream keeps the input embedding, so the next-token
    logit of succ(t) stands several sigma above the rest: greedy decoding walks the cycle (no fixed point, no near-ties)
    and token identity with the oracle can be demanded without a tie rule.  Returns a new weight dict."""
    t = cfg.text
    assert not t.tie_word_embeddings, "the peaked head needs an untied lm_head"
    n = n_cycle or t.vocab_size
    W = dict(W)
    E = W["language_model.model.embed_tokens.weight"].float()
    head = W["language_model.lm_head.weight"].float().clone()
    src = torch.arange(n)
    dst = (src + stride) % n
    head[dst] += gamma * E[src] / E[src].norm(dim=-1, keepdim=True).clamp_min(1e-6)
    W["language_model.lm_head.weight"] = head.to(W["language_model.model.embed_tokens.weight"].dtype)
    return W


def decode_teacher_forced(W, cfg: Cfg, input_ids, pixel_values=None, image_grid_thw=None, forced_tokens=(),
                          rope_mode: str = "fused", return_features: bool = False, kv_bits=None, kv_group_size: int = 64,
                          quantized_kv_start: int = 0, kv_batch_policy: bool = False, max_kv_size: Optional[int] = None):
    """generate_step's device work (generate/ar.py:334-389) with the FED tokens prescribed: full-prompt prefill, then one
    decode forward per forced token at pos = cache offset + rope_delta (language.py:476-509).
    -> logits [1 + len(forced_tokens), V]: row 0 = last prompt row, row i = after feeding forced_tokens[i-1]."""
    input_ids = np.asarray(input_ids)
    assert input_ids.shape[0] == 1
    emb, pos, deltas = get_input_embeddings(W, cfg, input_ids, pixel_values, image_grid_thw)
    cache = _make_prompt_cache(cfg, max_kv_size)
    from . import quant
    h = qwen2_model(W, cfg, emb, cache, torch.from_numpy(np.asarray(pos)), rope_mode)
    quant.maybe_quantize_kv_cache(cache, quantized_kv_start, kv_group_size, kv_bits, kv_batch_policy)    # ar.py:362: after every forward
    rows = [lm_head(W, cfg, h[:, -1:, :])[0, 0]]
    delta = int(deltas[0, 0])
    for y in forced_tokens:
        e = embed_tokens(W, np.array([[int(y)]]))
        pid = torch.full((3, 1, 1), _cache_offset(cache[0]) + delta, dtype=torch.long)
        h = qwen2_model(W, cfg, e, cache, pid, rope_mode)
        quant.maybe_quantize_kv_cache(cache, quantized_kv_start, kv_group_size, kv_bits, kv_batch_policy)
        rows.append(lm_head(W, cfg, h)[0, -1])
    out = torch.stack(rows)
    return (out, emb) if return_features else out
