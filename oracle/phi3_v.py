"""Oracle for the Phi-3.5-vision (`phi3_v`) path - SURVEY §8f row 2 (TEST INFRASTRUCTURE, see oracle/__init__.py).

torch-CPU restatement of the reference files

    mlx_vlm/models/phi3_v/vision.py           CLIP ViT-L/14-336 tower (Conv2d patch embed, class token, learned positions,
                                              pre_layrnorm, pre-LN encoder, FastGELUMLP), HD transform + sub_GN / glb_GN
                                              separators + img_projection (207-265)
    mlx_vlm/models/phi3_v/phi3_v.py           Phi-3 decoder (fused qkv_proj / gate_up_proj, no biases, SuScaledRoPE, silu(gate) * up),
                                              get_input_embeddings (negative ids = image positions), lm_head
    mlx_vlm/models/rope_utils.py:96-189       SuScaledRoPE: x * T(scale) (a typed multiply), then mx.fast.rope with
                                              freqs = factor * base ** (2 i / d)
    mlx_vlm/models/phi3_v/processing_phi3_v.py:78-236  HD image transform (resize / pad to multiples of 336, global + tiles)

on the primitives of oracle/ops.py (same typed-graph rounding policy).  Weight names are the reference's module tree
(`model.embed_tokens`, `model.layers.N...`, `model.vision_embed_tokens...`, `lm_head`).  Pinned by
tests/test_oracle_ref_golden_phi3v.py against vectors produced by the reference's own files executed over
oracle/mlx_shim (tests/golden/make_golden_ref_phi3v.py).

The tower's width is not configurable in the reference (`VisionModel.CLIP_VIT_LARGE_PATCH14_336_CONFIG`, image_dim_out =
1024 and the 12 x 12 merged grid are literals, vision.py:182-206,233-253); the tiny test configuration therefore keeps
hidden 1024 / 16 heads / 336 px / patch 14 and shrinks only depth and the MLP.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops

F32 = torch.float32

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class VisionCfg:
    """CLIP ViT-L/14-336 (vision.py:182-192)"""
    num_hidden_layers: int = 24
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_attention_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid


@dataclass
class TextCfg:
    hidden_size: int = 3072
    num_hidden_layers: int = 32
    intermediate_size: int = 8192
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    rms_norm_eps: float = 1e-5
    vocab_size: int = 32064
    rope_theta: float = 10000.0
    max_position_embeddings: int = 131072
    original_max_position_embeddings: int = 4096
    short_factor: Optional[List[float]] = None       # rope_scaling["short_factor"], head_dim / 2 entries; None = plain RoPE
    long_factor: Optional[List[float]] = None

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass
class Cfg:
    text: TextCfg = field(default_factory=TextCfg)
    vision: VisionCfg = field(default_factory=VisionCfg)


def su_factors(head_dim: int, seed: int = 7):
    """seeded stand-ins for the checkpoint's short / long factor lists (Phi-3.5's are 48 values in [1, 1.3] / [1, 65])"""
    rng = np.random.default_rng(seed)
    n = head_dim // 2
    short = np.sort(1.0 + 0.3 * rng.random(n)).round(4).tolist()
    long = np.sort(1.0 + 60.0 * rng.random(n)).round(4).tolist()
    return short, long


def tiny_cfg() -> Cfg:
    """Real head dims (96 text, 64 vision) and the literal CLIP width / grid of the reference's tower at toy depth."""
    short, long = su_factors(96)
    return Cfg(text=TextCfg(hidden_size=384, num_hidden_layers=2, intermediate_size=256, num_attention_heads=4,
                            num_key_value_heads=4, vocab_size=1024, short_factor=short, long_factor=long),
               vision=VisionCfg(num_hidden_layers=3, intermediate_size=256))


TEST_WEIGHT_SCALES = dict(std=0.1, embed_std=0.05)

M = "model."
VT = "model.vision_embed_tokens."
CLIP = VT + "img_processor.vision_model."


def random_weights(cfg: Cfg, seed: int = 0, dtype=torch.bfloat16, std: float = 0.05, embed_std: float = 0.2,
                   fast: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded weights under the reference's names (patch weight (O, kH, kW, C), as `VisionModel.sanitize` leaves it)."""
    g = torch.Generator().manual_seed(seed)
    v, t = cfg.vision, cfg.text

    n_fast = [0]

    def rn(*shape, s=std):
        if fast and math.prod(shape) >= (1 << 20):      # big matrices: threaded Philox streams (ops.fast_normal)
            n_fast[0] += 1
            return ops.fast_normal(shape, (seed, n_fast[0]), s, dtype)
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    def ln(prefix, dim):
        return {prefix + ".weight": (1 + 0.1 * torch.randn(dim, generator=g)).to(dtype), prefix + ".bias": rn(dim, s=0.1)}

    W: Dict[str, torch.Tensor] = {}
    E, I = v.hidden_size, v.intermediate_size
    ws = std * (64.0 / E) ** 0.5 * 2                         # keep the 1024-wide tower's activations O(1)
    W[CLIP + "embeddings.class_embedding"] = rn(E, s=0.3)
    W[CLIP + "embeddings.patch_embedding.weight"] = rn(E, v.patch_size, v.patch_size, v.num_channels, s=0.05)
    W[CLIP + "embeddings.position_embedding.weight"] = rn(v.num_patches + 1, E, s=0.3)
    W.update(ln(CLIP + "pre_layrnorm", E))
    for i in range(v.num_hidden_layers):
        p = f"{CLIP}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[p + f"self_attn.{n}.weight"] = rn(E, E, s=ws)
            W[p + f"self_attn.{n}.bias"] = rn(E, s=0.1)
        W.update(ln(p + "layer_norm1", E))
        W.update(ln(p + "layer_norm2", E))
        W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"] = rn(I, E, s=ws), rn(I, s=0.1)
        W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"] = rn(E, I, s=std), rn(E, s=0.1)
    W.update(ln(CLIP + "post_layernorm", E))
    D = t.hidden_size
    W[VT + "glb_GN"] = rn(1, 1, 4 * E, s=0.3)
    W[VT + "sub_GN"] = rn(1, 1, 1, 4 * E, s=0.3)
    W[VT + "img_projection.0.weight"], W[VT + "img_projection.0.bias"] = rn(D, 4 * E, s=ws / 2), rn(D, s=0.1)
    W[VT + "img_projection.2.weight"], W[VT + "img_projection.2.bias"] = rn(D, D), rn(D, s=0.1)
    hd = t.head_dim
    W[M + "embed_tokens.weight"] = rn(t.vocab_size, D, s=embed_std)
    for i in range(t.num_hidden_layers):
        p = f"{M}layers.{i}."
        W[p + "self_attn.qkv_proj.weight"] = rn((t.num_attention_heads + 2 * t.num_key_value_heads) * hd, D)
        W[p + "self_attn.o_proj.weight"] = rn(D, t.num_attention_heads * hd)
        W[p + "mlp.gate_up_proj.weight"] = rn(2 * t.intermediate_size, D)
        W[p + "mlp.down_proj.weight"] = rn(D, t.intermediate_size)
        W[p + "input_layernorm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype)
        W[p + "post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype)
    W[M + "norm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype)
    W["lm_head.weight"] = rn(t.vocab_size, D)
    return W


# --------------------------------------------------------------------------------------------- image processor
def calc_hd_transform_size(width: int, height: int, hd_num: int = 4):
    """_calc_hd_transform_size + _calc_padded_size (processing_phi3_v.py:78-110) -> (padded_width, padded_height)"""
    transposed = False
    if width < height:
        width, height = height, width
        transposed = True
    ratio = width / height
    scale = 1
    while scale * math.ceil(scale / ratio) <= hd_num:
        scale += 1
    scale -= 1
    new_width = int(scale * 336)
    new_height = int(new_width / ratio)
    pw, ph = math.ceil(new_width / 336) * 336, math.ceil(new_height / 336) * 336
    return (ph, pw) if transposed else (pw, ph)


def num_image_tokens(width: int, height: int, num_crops: int = 4, num_img_tokens: int = 144) -> int:
    """Phi3VImageProcessor.calc_num_image_tokens (processing_phi3_v.py:160-173)"""
    w, h = calc_hd_transform_size(width, height, num_crops)
    nh, nw = h // 336, w // 336
    return (nh * nw + 1) * num_img_tokens + 1 + (nh + 1) * 12


def preprocess(images_hwc_u8: Sequence[np.ndarray], num_crops: int = 4):
    """Phi3VImageProcessor.preprocess (processing_phi3_v.py:175-277): per image RGB -> resize to the HD size (PIL
    bicubic; the size is already a multiple of 336, so the pad is a no-op) -> global view = resize to 336 x 336 -> tiles
    row-major -> [global] + tiles, x / 255 (float32), (x - mean) / std in float64 (numpy promotes: the constants are
    float64 arrays), channels first; images with fewer tiles are zero-padded to the batch maximum.
    `mx.array(...)` of the float64 result is float32 (MLX has no float64 arrays on the device path): one rounding at the end.
    -> (pixel_values float32 [B, T, 3, 336, 336], image_sizes int [B, 2] = (hd_height, hd_width))"""
    from PIL import Image

    pvs, sizes = [], []
    for img in images_hwc_u8:
        pil = Image.fromarray(np.asarray(img, dtype=np.uint8)).convert("RGB")
        tw, th = calc_hd_transform_size(pil.size[0], pil.size[1], num_crops)
        hd = pil.resize((tw, th), Image.Resampling.BICUBIC)
        glb = hd.resize((336, 336), Image.Resampling.BICUBIC)
        tiles = [hd.crop((w * 336, h * 336, w * 336 + 336, h * 336 + 336)) for h in range(th // 336) for w in range(tw // 336)]
        proc = []
        for im in [glb] + tiles:
            arr = np.array(im, dtype=np.float32) / 255.0
            arr = (arr - np.array(OPENAI_CLIP_MEAN)) / np.array(OPENAI_CLIP_STD)
            proc.append(arr.transpose(2, 0, 1))
        pvs.append(np.stack(proc, axis=0))
        sizes.append((th, tw))
    T = max(p.shape[0] for p in pvs)
    pvs = [np.concatenate([p, np.zeros((T - p.shape[0], *p.shape[1:]), dtype=p.dtype)], 0) if p.shape[0] < T else p for p in pvs]
    return np.stack(pvs, axis=0).astype(np.float32), np.array(sizes)


# --------------------------------------------------------------------------------------------- CLIP tower
def vision_embeddings(W, cfg: Cfg, pixel_values: torch.Tensor) -> torch.Tensor:
    """VisionEmbeddings (vision.py:113-148): Conv2d(k = s = 14, no bias) over NHWC == one GEMM per patch with the patch
    flattened (kH, kW, C)-major; class embedding row in front; `+= position_embedding` (a typed add).
    pixel_values [N, 3, 336, 336] -> [N, 577, E]"""
    v = cfg.vision
    N = pixel_values.shape[0]
    P, G = v.patch_size, v.grid
    w = W[CLIP + "embeddings.patch_embedding.weight"]
    x = pixel_values.permute(0, 2, 3, 1)
    x = x.reshape(N, G, P, G, P, v.num_channels).permute(0, 1, 3, 2, 4, 5).reshape(N, G * G, P * P * v.num_channels)
    y = ops.linear(x, w.reshape(w.shape[0], -1))
    cls = W[CLIP + "embeddings.class_embedding"].to(y.dtype)[None, None].expand(N, 1, -1)
    return ops.add(torch.cat([cls, y], dim=1), W[CLIP + "embeddings.position_embedding.weight"][None])


def encoder_layer(W, i: int, cfg: Cfg, x: torch.Tensor) -> torch.Tensor:
    """EncoderLayer (vision.py:83-102) with Attention (28-80: separate q / k / v / out projections with bias, unmasked
    SDPA, scale head_dim ** -0.5) and FastGELUMLP (mlp.py:47-57)."""
    v = cfg.vision
    p = f"{CLIP}encoder.layers.{i}."
    B, L, E = x.shape
    H = v.num_attention_heads
    y = ops.layer_norm(x, W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], v.layer_norm_eps)
    q, k, vv = (ops.linear(y, W[p + f"self_attn.{n}.weight"], W[p + f"self_attn.{n}.bias"])
                .reshape(B, L, H, E // H).permute(0, 2, 1, 3) for n in ("q_proj", "k_proj", "v_proj"))
    o = ops.sdpa(q, k, vv, scale=(E // H) ** -0.5).permute(0, 2, 1, 3).reshape(B, L, E)
    x = ops.add(x, ops.linear(o, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"]))
    y = ops.layer_norm(x, W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], v.layer_norm_eps)
    y = ops.gelu_fast(ops.linear(y, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"]))
    return ops.add(x, ops.linear(y, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"]))


def clip_features(W, cfg: Cfg, pixel_values: torch.Tensor, embeddings=None, return_states: bool = False):
    """ClipModel (vision.py:151-176) as the HD transform uses it (vision.py:224-226): `encoder_states[-2][:, 1:]` - the
    state after all layers but the last, class row dropped; nn.LayerNorm(hidden) `pre_layrnorm` has MLX's default eps
    1e-5.  The last layer and post_layernorm cannot influence the output and are not computed.  -> [N, 576, E]"""
    x = vision_embeddings(W, cfg, pixel_values) if embeddings is None else embeddings
    x = ops.layer_norm(x, W[CLIP + "pre_layrnorm.weight"], W[CLIP + "pre_layrnorm.bias"], 1e-5)
    states = [x]
    for i in range(cfg.vision.num_hidden_layers - 1):
        x = encoder_layer(W, i, cfg, x)
        states.append(x)
    return (x[:, 1:], states) if return_states else x[:, 1:]


def hd_rows(feat: torch.Tensor, h: int, w: int, sub_gn: torch.Tensor, glb_gn: torch.Tensor) -> torch.Tensor:
    """The row assembly of vision.py:233-256 for one image.  feat [1 + h w (+ padding), 576, C]: view 0 is the global
    image.  A view's 24 x 24 grid is merged 2 x 2 into 12 x 12 rows of 4 C ((a, b, i, j) -> row (a, b), columns (i, j, c));
    the h w local views are laid out by a PLAIN reshape of [h w, 12, 12, 4 C] to [h 12, w 12, 4 C] (the reference's
    own arrangement, not a tile mosaic); every row of a grid gets one sub_GN column appended; order: local rows, glb_GN,
    global rows.  -> [(h w + 1) 144 + 1 + (h + 1) 12, 4 C]"""
    C = feat.shape[-1]
    Hh = int(round(feat.shape[1] ** 0.5)) // 2

    def merged(x):          # [n, 576, C] -> [n, 12, 12, 4 C]
        n = x.shape[0]
        return x.reshape(n, Hh, 2, Hh, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(n, Hh, Hh, 4 * C)

    def with_sep(grid):     # [1, R, S, 4 C] -> [R (S + 1), 4 C]
        R = grid.shape[1]
        sep = sub_gn.to(grid.dtype).reshape(1, 1, 1, 4 * C).expand(1, R, 1, 4 * C)
        return torch.cat([grid, sep], dim=2).reshape(-1, 4 * C)

    glb = with_sep(merged(feat[:1]))
    sub = with_sep(merged(feat[1: 1 + h * w]).reshape(1, h * Hh, w * Hh, 4 * C))
    return torch.cat([sub, glb_gn.to(feat.dtype).reshape(1, 4 * C), glb], dim=0)


def img_projection(W, x: torch.Tensor) -> torch.Tensor:
    """img_projection (vision.py:200-204): Linear -> nn.GELU() (erf) -> Linear"""
    h = ops.gelu_erf(ops.linear(x, W[VT + "img_projection.0.weight"], W[VT + "img_projection.0.bias"]))
    return ops.linear(h, W[VT + "img_projection.2.weight"], W[VT + "img_projection.2.bias"])


def image_features(W, cfg: Cfg, pixel_values: torch.Tensor, image_sizes, clip_embeddings=None) -> List[torch.Tensor]:
    """VisionModel.__call__ (vision.py:207-262) up to the projected rows of every image.  pixel_values [B, T, 3, 336, 336],
    image_sizes [B, 2] (height, width in pixels).  -> list of [cnt_b, hidden]"""
    B, T = pixel_values.shape[:2]
    feat = clip_features(W, cfg, pixel_values.reshape(B * T, *pixel_values.shape[2:]), embeddings=clip_embeddings)
    feat = feat.reshape(B, T, *feat.shape[1:])
    out = []
    for b in range(B):
        h, w = (int(image_sizes[b][0]) // 336, int(image_sizes[b][1]) // 336)
        out.append(img_projection(W, hd_rows(feat[b], h, w, W[VT + "sub_GN"], W[VT + "glb_GN"])))
    return out


# --------------------------------------------------------------------------------------------- language model
def embed_tokens(W, input_ids) -> torch.Tensor:
    """nn.Embedding on the ids as given: negative ids (image positions) wrap like python indices; those rows are
    overwritten by image features afterwards"""
    w, idx = W[M + "embed_tokens.weight"], torch.as_tensor(np.asarray(input_ids), dtype=torch.long)
    return w.rows(idx) if hasattr(w, "wq") else w[idx]          # nn.QuantizedEmbedding: mx.dequantize of the gathered rows


def get_input_embeddings(W, cfg: Cfg, input_ids, pixel_values: Optional[torch.Tensor] = None, image_sizes=None,
                         clip_embeddings=None) -> torch.Tensor:
    """Model.get_input_embeddings (phi3_v.py:199-233) + the write-back of vision.py:257-262: the run of negative ids of
    image i starts at the (sum of earlier counts)-th negative position; cnt rows are written from there."""
    ids = np.asarray(input_ids)
    emb = embed_tokens(W, ids).clone()
    if pixel_values is None:
        return emb
    pix = pixel_values.to(emb.dtype)
    positions = np.argwhere(ids < 0).tolist()
    idx = 0
    for rows in image_features(W, cfg, pix, image_sizes, clip_embeddings):
        b, start = positions[idx]
        cnt = rows.shape[0]
        emb[b, start:start + cnt] = rows.to(emb.dtype)
        idx += cnt
    return emb


def su_rope_tables(t: TextCfg, dtype=torch.bfloat16):
    """SuScaledRoPE.__init__ (rope_utils.py:131-153) -> (short inv_freq, long inv_freq, scale as the model dtype sees it)"""
    hd = t.head_dim
    freqs = t.rope_theta ** (torch.arange(0, hd, 2, dtype=F32) / hd)
    if t.short_factor is None:
        return 1.0 / freqs, 1.0 / freqs, 1.0
    factor = t.max_position_embeddings / t.original_max_position_embeddings
    scale = 1.0 if factor <= 1.0 else math.sqrt(1 + math.log(factor) / math.log(t.original_max_position_embeddings))
    s = float(torch.tensor(scale, dtype=F32).to(dtype).to(F32))
    return (1.0 / (torch.tensor(t.short_factor, dtype=F32) * freqs), 1.0 / (torch.tensor(t.long_factor, dtype=F32) * freqs), s)


def su_rope(x: torch.Tensor, offset: int, t: TextCfg, position_end: Optional[int] = None) -> torch.Tensor:
    """SuScaledRoPE.__call__ (rope_utils.py:168-189): position_end = max(offset over the rows of the CALL) + L; long
    factors (and the long scale) for EVERY row of the call iff position_end > original_max_position_embeddings; x *
    T(scale) is a typed multiply; mx.fast.rope then rotates half-split pairs with fp32 angles pos / freqs and one
    rounding.  This function rotates ONE sequence at cache offset `offset`; `position_end` carries the call-wide value
    when the sequence is a row of a batched call (None: the sequence is the whole call, position_end = offset + L)."""
    L = x.shape[-2]
    inv_s, inv_l, s = su_rope_tables(t, x.dtype)
    pe = offset + L if position_end is None else int(position_end)
    inv = inv_l if pe > t.original_max_position_embeddings else inv_s
    if t.short_factor is not None:
        x = (x.to(F32) * s).to(x.dtype)
    pos = torch.arange(offset, offset + L)[None].expand(x.shape[0], L)
    return ops.mrope_apply(x, pos, inv, None, "fused")


def attention(W, p: str, cfg: Cfg, x: torch.Tensor, cache: Optional[ops.KVCache], position_end: Optional[int] = None) -> torch.Tensor:
    """Attention (phi3_v.py:17-94): one qkv Linear (no bias) split [q | k | v], rope at cache.offset, KVCache, causal
    SDPA at head_dim ** -0.5, o_proj."""
    t = cfg.text
    B, L, D = x.shape
    H, Hkv, hd = t.num_attention_heads, t.num_key_value_heads, t.head_dim
    qkv = ops.linear(x, W[p + "qkv_proj.weight"])
    q = qkv[..., : H * hd].reshape(B, L, H, hd).permute(0, 2, 1, 3)
    k = qkv[..., H * hd: (H + Hkv) * hd].reshape(B, L, Hkv, hd).permute(0, 2, 1, 3)
    v = qkv[..., (H + Hkv) * hd:].reshape(B, L, Hkv, hd).permute(0, 2, 1, 3)
    off = cache.offset if cache is not None else 0
    q, k = su_rope(q, off, t, position_end), su_rope(k, off, t, position_end)
    if cache is not None:
        k, v = cache.update_and_fetch(k, v)
    o = ops.sdpa(q, k, v, scale=hd ** -0.5, causal=L > 1, q_offset=k.shape[2] - L)
    return ops.linear(o.permute(0, 2, 1, 3).reshape(B, L, -1), W[p + "o_proj.weight"])


def decoder_layer(W, i: int, cfg: Cfg, x: torch.Tensor, cache, position_end: Optional[int] = None) -> torch.Tensor:
    """TransformerBlock (phi3_v.py:109-133) with MLP (96-106): gate_up Linear split [gate | up], silu(gate) * up, down."""
    p = f"{M}layers.{i}."
    eps = cfg.text.rms_norm_eps
    h = ops.add(x, attention(W, p + "self_attn.", cfg, ops.rms_norm(x, W[p + "input_layernorm.weight"], eps), cache, position_end))
    gu = ops.linear(ops.rms_norm(h, W[p + "post_attention_layernorm.weight"], eps), W[p + "mlp.gate_up_proj.weight"])
    I = gu.shape[-1] // 2
    return ops.add(h, ops.linear(ops.swiglu(gu[..., :I], gu[..., I:]), W[p + "mlp.down_proj.weight"]))


def language_model(W, cfg: Cfg, inputs_embeds: torch.Tensor, cache=None, last_only: bool = False,
                   position_end: Optional[int] = None) -> torch.Tensor:
    """Phi3V layers -> norm -> lm_head (phi3_v.py:136-197).  -> logits [B, L, V] (last_only: [B, 1, V], same values).
    position_end: see su_rope (the rope regime of a batched call is decided by its longest row)."""
    h = inputs_embeds
    cache = cache or [None] * cfg.text.num_hidden_layers
    for i in range(cfg.text.num_hidden_layers):
        h = decoder_layer(W, i, cfg, h, cache[i], position_end)
    if last_only:
        h = h[:, -1:, :]
    return ops.linear(ops.rms_norm(h, W[M + "norm.weight"], cfg.text.rms_norm_eps), W["lm_head.weight"])


def generate_greedy(W, cfg: Cfg, input_ids, pixel_values=None, image_sizes=None, max_tokens: int = 8, return_logits: bool = False,
                    clip_embeddings=None):
    """generate_step, temperature 0 (generate/ar.py:151-515)"""
    ids = np.asarray(input_ids)
    assert ids.shape[0] == 1
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    emb = get_input_embeddings(W, cfg, ids, pixel_values, image_sizes, clip_embeddings)
    logits = language_model(W, cfg, emb, cache, last_only=True)[:, -1, :]
    toks, rows = [], []
    for n in range(max_tokens):
        y = int(ops.argmax_first(ops.logprobs_from_logits(logits))[0])
        toks.append(y)
        rows.append(logits[0].clone())
        if n == max_tokens - 1:
            break
        logits = language_model(W, cfg, embed_tokens(W, np.array([[y]])), cache)[:, -1, :]
    return (toks, torch.stack(rows)) if return_logits else toks


def decode_teacher_forced(W, cfg: Cfg, input_ids, pixel_values=None, image_sizes=None, forced_tokens=()) -> torch.Tensor:
    """Test construction (no reference counterpart): generate_step's device work with the FED tokens prescribed.
    -> logits [1 + len(forced_tokens), V]: row 0 = last prompt row of the prefill, row i = after feeding forced_tokens[i - 1]."""
    ids = np.asarray(input_ids)
    assert ids.shape[0] == 1
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    rows = [language_model(W, cfg, get_input_embeddings(W, cfg, ids, pixel_values, image_sizes), cache, last_only=True)[0, 0]]
    for y in forced_tokens:
        rows.append(language_model(W, cfg, embed_tokens(W, np.array([[int(y)]])), cache)[0, 0])
    return torch.stack(rows)
