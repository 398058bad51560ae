"""Oracle for the Idefics2 path - SURVEY §8f row 3 / BASELINE configs[3] (TEST INFRASTRUCTURE, see oracle/__init__.py).

torch-CPU restatement of the reference files

    mlx_vlm/models/idefics2/vision.py      SigLIP-style tower: Conv2d patch embed (+bias), position ids bucketed from the patch
                                           mask (129-172), pre-LN encoder WITHOUT an attention mask (the reference passes none to
                                           the encoder, 196-199), post_layernorm (MLX default eps 1e-5)
    mlx_vlm/models/idefics2/idefics2.py    connector = modality projection (silu-gated MLP) + perceiver resampler (64 latents,
                                           GQA cross-attention over [context | latents], 36-147), padding-image removal, patch
                                           mask, masked_scatter (15-33, 170-262), sanitize (294-321)
    mlx_vlm/models/idefics2/language.py    Mistral decoder (no biases, nn.RoPE, SwiGLU, untied head)
    transformers Idefics2ImageProcessor    (the reference uses HF's image processor, processing_idefics2.py:186-214): resize to
                                           shortest_edge 378 / longest_edge 980 bilinear, 1/255, (x - 0.5) / 0.5, pad to the batch
                                           maximum + pixel_attention_mask, optional 4 + 1 image splitting

on the primitives of oracle/ops.py.  Weight names are the reference's module tree after `Model.sanitize`.  Pinned by
tests/test_oracle_ref_golden_idefics2.py against vectors produced by the reference's own files executed over
oracle/mlx_shim (tests/golden/make_golden_ref_idefics2.py) and, for the image processor, by transformers' PIL backend.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import math
import numpy as np
import torch

from . import ops

F32 = torch.float32


@dataclass
class VisionCfg:
    hidden_size: int = 1152
    intermediate_size: int = 4304
    num_hidden_layers: int = 27
    num_attention_heads: int = 16
    num_channels: int = 3
    image_size: int = 980
    patch_size: int = 14
    layer_norm_eps: float = 1e-6

    @property
    def num_patches_per_side(self) -> int:
        return self.image_size // self.patch_size


@dataclass
class TextCfg:
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    rms_norm_eps: float = 1e-5
    vocab_size: int = 32003
    rope_theta: float = 10000.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass
class PerceiverCfg:
    num_key_value_heads: int = 4
    resampler_depth: int = 3
    resampler_head_dim: int = 96
    resampler_n_heads: int = 16
    resampler_n_latents: int = 64


@dataclass
class Cfg:
    text: TextCfg = field(default_factory=TextCfg)
    vision: VisionCfg = field(default_factory=VisionCfg)
    perceiver: PerceiverCfg = field(default_factory=PerceiverCfg)
    image_token_id: int = 32001


def tiny_cfg() -> Cfg:
    """Real head dims (72 vision, 128 text, 96 perceiver with GQA 4:1... 2:1 here), a 10 x 10 position table at toy widths."""
    return Cfg(text=TextCfg(hidden_size=256, intermediate_size=384, num_hidden_layers=2, num_attention_heads=2,
                            num_key_value_heads=1, vocab_size=1024),
               vision=VisionCfg(hidden_size=144, intermediate_size=288, num_hidden_layers=2, num_attention_heads=2, image_size=140),
               perceiver=PerceiverCfg(num_key_value_heads=1, resampler_depth=2, resampler_head_dim=96, resampler_n_heads=2,
                                      resampler_n_latents=8),
               image_token_id=1001)


TEST_WEIGHT_SCALES = dict(std=0.05, embed_std=0.2)

V = "vision_model."
C = "connector."
PR = "connector.perceiver_resampler."
LM = "language_model."


def random_weights(cfg: Cfg, seed: int = 0, dtype=torch.bfloat16, std: float = 0.05, embed_std: float = 0.2,
                   fast: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded weights under the reference's (sanitized) names; patch weight (O, kH, kW, C)."""
    g = torch.Generator().manual_seed(seed)
    v, t, p = cfg.vision, cfg.text, cfg.perceiver

    n_fast = [0]

    def rn(*shape, s=std):
        if fast and math.prod(shape) >= (1 << 20):      # big matrices: threaded Philox streams (ops.fast_normal)
            n_fast[0] += 1
            return ops.fast_normal(shape, (seed, n_fast[0]), s, dtype)
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    def nw(dim):
        return (1 + 0.1 * torch.randn(dim, generator=g)).to(dtype)

    def ln(prefix, dim):
        return {prefix + ".weight": nw(dim), prefix + ".bias": rn(dim, s=0.1)}

    W: Dict[str, torch.Tensor] = {}
    E, I = v.hidden_size, v.intermediate_size
    W[V + "embeddings.patch_embedding.weight"] = rn(E, v.patch_size, v.patch_size, v.num_channels)
    W[V + "embeddings.patch_embedding.bias"] = rn(E, s=0.1)
    W[V + "embeddings.position_embedding.weight"] = rn(v.num_patches_per_side ** 2, E, s=0.1)
    for i in range(v.num_hidden_layers):
        q = f"{V}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[q + f"self_attn.{n}.weight"], W[q + f"self_attn.{n}.bias"] = rn(E, E), rn(E, s=0.1)
        W.update(ln(q + "layer_norm1", E))
        W.update(ln(q + "layer_norm2", E))
        W[q + "mlp.fc1.weight"], W[q + "mlp.fc1.bias"] = rn(I, E), rn(I, s=0.1)
        W[q + "mlp.fc2.weight"], W[q + "mlp.fc2.bias"] = rn(E, I), rn(E, s=0.1)
    W.update(ln(V + "post_layernorm", E))
    D, TI = t.hidden_size, t.intermediate_size
    W[C + "modality_projection.gate_proj.weight"] = rn(TI, E)
    W[C + "modality_projection.up_proj.weight"] = rn(TI, E)
    W[C + "modality_projection.down_proj.weight"] = rn(D, TI)
    W[PR + "latents"] = (1 + 0.5 * torch.randn(p.resampler_n_latents, D, generator=g)).to(dtype)
    ph = p.resampler_head_dim
    for i in range(p.resampler_depth):
        q = f"{PR}layers.{i}."
        W[q + "input_latents_norm.weight"], W[q + "input_context_norm.weight"] = nw(D), nw(D)
        W[q + "self_attn.q_proj.weight"] = rn(p.resampler_n_heads * ph, D)
        W[q + "self_attn.k_proj.weight"] = rn(p.num_key_value_heads * ph, D)
        W[q + "self_attn.v_proj.weight"] = rn(p.num_key_value_heads * ph, D)
        W[q + "self_attn.o_proj.weight"] = rn(D, p.resampler_n_heads * ph)
        W[q + "post_attention_layernorm.weight"] = nw(D)
        W[q + "mlp.gate_proj.weight"], W[q + "mlp.up_proj.weight"] = rn(4 * D, D), rn(4 * D, D)
        W[q + "mlp.down_proj.weight"] = rn(D, 4 * D)
    W[PR + "norm.weight"] = nw(D)
    hd = t.head_dim
    W[LM + "embed_tokens.weight"] = rn(t.vocab_size, D, s=embed_std)
    for i in range(t.num_hidden_layers):
        q = f"{LM}layers.{i}."
        W[q + "self_attn.q_proj.weight"] = rn(t.num_attention_heads * hd, D)
        W[q + "self_attn.k_proj.weight"] = rn(t.num_key_value_heads * hd, D)
        W[q + "self_attn.v_proj.weight"] = rn(t.num_key_value_heads * hd, D)
        W[q + "self_attn.o_proj.weight"] = rn(D, t.num_attention_heads * hd)
        W[q + "mlp.gate_proj.weight"], W[q + "mlp.up_proj.weight"] = rn(TI, D), rn(TI, D)
        W[q + "mlp.down_proj.weight"] = rn(D, TI)
        W[q + "input_layernorm.weight"], W[q + "post_attention_layernorm.weight"] = nw(D), nw(D)
    W[LM + "norm.weight"] = nw(D)
    W[LM + "lm_head.weight"] = rn(t.vocab_size, D)
    return W


# --------------------------------------------------------------------------------------------- image processor
def resize_output_size(height: int, width: int, shortest_edge: int = 378, longest_edge: int = 980):
    """transformers get_resize_output_image_size (idefics2): cap the longer side at longest_edge (aspect kept, int()), then
    raise both sides to at least shortest_edge"""
    aspect = width / height
    if width >= height and width > longest_edge:
        width = longest_edge
        height = int(width / aspect)
    elif height > width and height > longest_edge:
        height = longest_edge
        width = int(height * aspect)
    return max(height, shortest_edge), max(width, shortest_edge)


def preprocess(images: Sequence[Sequence[np.ndarray]], shortest_edge: int = 378, longest_edge: int = 980,
               do_image_splitting: bool = False):
    """Idefics2ImageProcessor: per image RGB -> (4 quadrants + the image when splitting) -> bilinear resize (PIL) -> x * (1 /
    255) -> (x - 0.5) / 0.5 -> channels first; all images zero-padded (bottom / right) to the largest, samples with fewer
    images padded with all-zero images.  images: one list of HWC uint8 arrays per sample.
    -> (pixel_values float32 [B, N, 3, H, W], pixel_attention_mask int64 [B, N, H, W])"""
    from PIL import Image

    out = []
    for sample in images:
        row = []
        for img in sample:
            a = np.asarray(img, dtype=np.uint8)
            parts = [a]
            if do_image_splitting:
                mh, mw = a.shape[0] // 2, a.shape[1] // 2
                parts = [a[:mh, :mw], a[:mh, mw:], a[mh:, :mw], a[mh:, mw:], a]
            for part in parts:
                h, w = resize_output_size(part.shape[0], part.shape[1], shortest_edge, longest_edge)
                pil = Image.fromarray(part).convert("RGB").resize((w, h), resample=Image.BILINEAR)
                x = (np.asarray(pil).astype(np.float64) * (1 / 255)).astype(np.float32)      # transformers.rescale
                x = (x - np.float32(0.5)) / np.float32(0.5)
                row.append(np.transpose(x, (2, 0, 1)))
        out.append(row)
    N = max(len(r) for r in out)
    H = max(x.shape[1] for r in out for x in r)
    Wd = max(x.shape[2] for r in out for x in r)
    pv = np.zeros((len(out), N, 3, H, Wd), dtype=np.float32)
    mask = np.zeros((len(out), N, H, Wd), dtype=np.int64)
    for i, r in enumerate(out):
        for j, x in enumerate(r):
            pv[i, j, :, : x.shape[1], : x.shape[2]] = x
            mask[i, j, : x.shape[1], : x.shape[2]] = 1
    return pv, mask


# --------------------------------------------------------------------------------------------- vision tower
def real_images_and_patch_mask(pixel_values: torch.Tensor, pixel_attention_mask, patch_size: int):
    """Model.get_input_embeddings (idefics2.py:184-226): [B, N, C, H, W] -> the non-padding images [n, C, H, W] (a padding
    image is all zeros) and their patch masks bool [n, H / p, W / p] (a patch is live iff any of its pixels is)."""
    B, N, Cc, H, Wd = pixel_values.shape
    pv = pixel_values.reshape(B * N, Cc, H, Wd)
    real = ((pv == 0.0).sum(dim=(-1, -2, -3)) != Cc * H * Wd).numpy()
    inds = np.where(real)[0].tolist()
    pv = pv[inds]
    if pixel_attention_mask is None:
        pam = np.ones((len(inds), H, Wd), dtype=bool)
    else:
        pam = np.asarray(pixel_attention_mask).reshape(B * N, H, Wd)[inds] > 0
    ph, pw = H // patch_size, Wd // patch_size
    r = pam[:, : ph * patch_size, : pw * patch_size].reshape(len(inds), ph, patch_size, pw, patch_size)
    return pv, r.transpose(0, 1, 3, 2, 4).sum(axis=(-1, -2)) > 0


def position_ids(patch_mask: np.ndarray, num_patches_per_side: int) -> np.ndarray:
    """VisionEmbeddings.__call__ (vision.py:143-166): fractional coordinates of the LIVE patch rows / columns bucketed into the
    num_patches_per_side grid of the position table; dead patches keep id 0.  patch_mask bool [n, gh, gw] -> int [n, gh gw]"""
    n, gh, gw = patch_mask.shape
    S = num_patches_per_side
    boundaries = np.linspace(1 / S, 1.0, S, endpoint=False)
    ids = np.zeros((n, gh * gw), dtype=np.int64)
    for b in range(n):
        m = patch_mask[b]
        nh, nw = int(m[:, 0].sum()), int(m[0, :].sum())
        bh = np.digitize(np.linspace(0, 1, nh, endpoint=False), boundaries, right=True) - 1
        bw = np.digitize(np.linspace(0, 1, nw, endpoint=False), boundaries, right=True) - 1
        ids[b][m.reshape(-1)] = (bh[:, None] * S + bw).flatten()
    return ids


def vision_embeddings(W, cfg: Cfg, pixel_values: torch.Tensor, patch_mask: np.ndarray) -> torch.Tensor:
    """Conv2d(k = s = patch, bias) over NHWC == one GEMM per patch flattened (kH, kW, C)-major, `+= position_embedding(ids)`
    (a typed add).  pixel_values [n, 3, H, W] -> [n, gh gw, E]"""
    v = cfg.vision
    n, Cc, H, Wd = pixel_values.shape
    P = v.patch_size
    gh, gw = H // P, Wd // P
    w = W[V + "embeddings.patch_embedding.weight"]
    x = pixel_values.permute(0, 2, 3, 1)[:, : gh * P, : gw * P]
    x = x.reshape(n, gh, P, gw, P, Cc).permute(0, 1, 3, 2, 4, 5).reshape(n, gh * gw, P * P * Cc)
    y = ops.linear(x, w.reshape(w.shape[0], -1), W[V + "embeddings.patch_embedding.bias"])
    ids = torch.from_numpy(position_ids(patch_mask, v.num_patches_per_side))
    return ops.add(y, W[V + "embeddings.position_embedding.weight"][ids])


def encoder_layer(W, i: int, cfg: Cfg, x: torch.Tensor) -> torch.Tensor:
    """EncoderLayer (vision.py:84-99): pre-LN, unmasked attention over all patches of an image (padding patches included -
    the reference hands no mask to the encoder), FastGELUMLP."""
    v = cfg.vision
    p = f"{V}encoder.layers.{i}."
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


def vision_tower(W, cfg: Cfg, pixel_values: torch.Tensor, patch_mask: np.ndarray, embeddings=None, return_states: bool = False):
    """VisionModel.__call__ (vision.py:186-205) -> pooler_output = post_layernorm(last encoder state) [n, L, E]"""
    x = vision_embeddings(W, cfg, pixel_values, patch_mask) if embeddings is None else embeddings
    states = [x]
    for i in range(cfg.vision.num_hidden_layers):
        x = encoder_layer(W, i, cfg, x)
        states.append(x)
    out = ops.layer_norm(x, W[V + "post_layernorm.weight"], W[V + "post_layernorm.bias"], 1e-5)
    return (out, states) if return_states else out


# --------------------------------------------------------------------------------------------- connector
def gated_mlp(W, p: str, x: torch.Tensor) -> torch.Tensor:
    """MLP (idefics2.py:150-160): down(silu(gate(x)) * up(x))"""
    return ops.linear(ops.swiglu(ops.linear(x, W[p + "gate_proj.weight"]), ops.linear(x, W[p + "up_proj.weight"])), W[p + "down_proj.weight"])


def perceiver_layer(W, i: int, cfg: Cfg, latents: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
    """Idefics2PerceiverLayer + Attention (idefics2.py:36-123): queries from the normed latents, keys / values from
    [normed context | normed latents], GQA, no mask, no rope."""
    p, eps = cfg.perceiver, cfg.text.rms_norm_eps
    q_ = f"{PR}layers.{i}."
    B, L, D = latents.shape
    H, Hkv, hd = p.resampler_n_heads, p.num_key_value_heads, p.resampler_head_dim
    lat = ops.rms_norm(latents, W[q_ + "input_latents_norm.weight"], eps)
    ctx = ops.rms_norm(context, W[q_ + "input_context_norm.weight"], eps)
    hs = torch.cat([ctx, lat], dim=-2)
    S = hs.shape[1]
    q = ops.linear(lat, W[q_ + "self_attn.q_proj.weight"]).reshape(B, L, H, hd).permute(0, 2, 1, 3)
    k = ops.linear(hs, W[q_ + "self_attn.k_proj.weight"]).reshape(B, S, Hkv, hd).permute(0, 2, 1, 3)
    v = ops.linear(hs, W[q_ + "self_attn.v_proj.weight"]).reshape(B, S, Hkv, hd).permute(0, 2, 1, 3)
    o = ops.sdpa(q, k, v, scale=hd ** -0.5).permute(0, 2, 1, 3).reshape(B, L, -1)
    lat2 = ops.add(latents, ops.linear(o, W[q_ + "self_attn.o_proj.weight"]))
    return ops.add(lat2, gated_mlp(W, q_ + "mlp.", ops.rms_norm(lat2, W[q_ + "post_attention_layernorm.weight"], eps)))


def connector(W, cfg: Cfg, image_hidden: torch.Tensor) -> torch.Tensor:
    """Idefics2Connector (idefics2.py:163-177): modality projection, then the resampler from the learned latents.
    [n, L, E] -> [n, n_latents, D]"""
    x = gated_mlp(W, C + "modality_projection.", image_hidden)
    h = W[PR + "latents"].to(x.dtype)[None].expand(x.shape[0], -1, -1)
    for i in range(cfg.perceiver.resampler_depth):
        h = perceiver_layer(W, i, cfg, h, x)
    return ops.rms_norm(h, W[PR + "norm.weight"], cfg.text.rms_norm_eps)


def image_features(W, cfg: Cfg, pixel_values: torch.Tensor, pixel_attention_mask=None, embeddings=None) -> torch.Tensor:
    """-> [n_real_images, n_latents, D]"""
    pv, pmask = real_images_and_patch_mask(pixel_values, pixel_attention_mask, cfg.vision.patch_size)
    pooled = vision_tower(W, cfg, pv, pmask, embeddings=embeddings)
    return connector(W, cfg, pooled.to(pv.dtype))


# --------------------------------------------------------------------------------------------- language model
def embed_tokens(W, input_ids) -> torch.Tensor:
    return W[LM + "embed_tokens.weight"][torch.as_tensor(np.asarray(input_ids), dtype=torch.long)]


def get_input_embeddings(W, cfg: Cfg, input_ids, pixel_values: Optional[torch.Tensor] = None, pixel_attention_mask=None,
                         vision_embeddings_override=None, cast_pixels: bool = True) -> torch.Tensor:
    """Model.get_input_embeddings + masked_scatter (idefics2.py:15-33, 170-262): the rows at `image_token_id` take the
    resampler outputs in order.  cast_pixels=False is the reference AS SHIPPED: like llava_bunny it never casts the pixels
    to the weight dtype, so float32 pixels (what prepare_inputs hands over) promote the tower and the connector to float32
    activations; the scatter into the bf16 prompt rounds them once at the end.  True = the bf16 typed graph."""
    ids = np.asarray(input_ids)
    emb = embed_tokens(W, ids).clone()
    if pixel_values is None:
        return emb
    pix = pixel_values.to(emb.dtype) if cast_pixels else pixel_values
    feats = image_features(W, cfg, pix, pixel_attention_mask, vision_embeddings_override)
    flat = feats.reshape(-1, feats.shape[-1]).to(emb.dtype)
    where = np.argwhere(ids == cfg.image_token_id)
    if len(where) != flat.shape[0]:
        raise ValueError(f"Image features and image tokens do not match: tokens: {len(where)}, features {flat.shape[0]}")
    emb[torch.as_tensor(where[:, 0]), torch.as_tensor(where[:, 1])] = flat
    return emb


def attention(W, p: str, cfg: Cfg, x: torch.Tensor, cache: Optional[ops.KVCache]) -> torch.Tensor:
    """Attention (language.py:15-70): bias-free q / k / v, nn.RoPE(head_dim, base=rope_theta) at the cache offset, KVCache,
    causal SDPA, o_proj."""
    t = cfg.text
    B, L, D = x.shape
    H, Hkv, hd = t.num_attention_heads, t.num_key_value_heads, t.head_dim
    q = ops.linear(x, W[p + "q_proj.weight"]).reshape(B, L, H, hd).permute(0, 2, 1, 3)
    k = ops.linear(x, W[p + "k_proj.weight"]).reshape(B, L, Hkv, hd).permute(0, 2, 1, 3)
    v = ops.linear(x, W[p + "v_proj.weight"]).reshape(B, L, Hkv, hd).permute(0, 2, 1, 3)
    off = cache.offset if cache is not None else 0
    pos = torch.arange(off, off + L)[None].expand(B, L)
    inv = ops.mrope_inv_freq(hd, t.rope_theta)
    q, k = ops.mrope_apply(q, pos, inv, None, "fused"), ops.mrope_apply(k, pos, inv, None, "fused")
    if cache is not None:
        k, v = cache.update_and_fetch(k, v)
    o = ops.sdpa(q, k, v, scale=hd ** -0.5, causal=L > 1, q_offset=k.shape[2] - L)
    return ops.linear(o.permute(0, 2, 1, 3).reshape(B, L, -1), W[p + "o_proj.weight"])


def decoder_layer(W, i: int, cfg: Cfg, x: torch.Tensor, cache) -> torch.Tensor:
    p = f"{LM}layers.{i}."
    eps = cfg.text.rms_norm_eps
    h = ops.add(x, attention(W, p + "self_attn.", cfg, ops.rms_norm(x, W[p + "input_layernorm.weight"], eps), cache))
    return ops.add(h, gated_mlp(W, p + "mlp.", ops.rms_norm(h, W[p + "post_attention_layernorm.weight"], eps)))


def language_model(W, cfg: Cfg, inputs_embeds: torch.Tensor, cache=None, last_only: bool = False) -> torch.Tensor:
    h = inputs_embeds
    cache = cache or [None] * cfg.text.num_hidden_layers
    for i in range(cfg.text.num_hidden_layers):
        h = decoder_layer(W, i, cfg, h, cache[i])
    if last_only:
        h = h[:, -1:, :]
    return ops.linear(ops.rms_norm(h, W[LM + "norm.weight"], cfg.text.rms_norm_eps), W[LM + "lm_head.weight"])


def generate_greedy(W, cfg: Cfg, input_ids, pixel_values=None, pixel_attention_mask=None, max_tokens: int = 8,
                    return_logits: bool = False, vision_embeddings_override=None):
    ids = np.asarray(input_ids)
    assert ids.shape[0] == 1
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    emb = get_input_embeddings(W, cfg, ids, pixel_values, pixel_attention_mask, vision_embeddings_override)
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


def decode_teacher_forced(W, cfg: Cfg, input_ids, pixel_values=None, pixel_attention_mask=None, forced_tokens=()) -> torch.Tensor:
    """Test construction: generate_step's device work with the FED tokens prescribed -> logits [1 + len(forced), V]"""
    ids = np.asarray(input_ids)
    assert ids.shape[0] == 1
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    rows = [language_model(W, cfg, get_input_embeddings(W, cfg, ids, pixel_values, pixel_attention_mask), cache, last_only=True)[0, 0]]
    for y in forced_tokens:
        rows.append(language_model(W, cfg, embed_tokens(W, np.array([[int(y)]])), cache)[0, 0])
    return torch.stack(rows)
