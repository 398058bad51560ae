"""Oracle for the nanoLLaVA (`llava_bunny`) path - SURVEY §8f row 1 (TEST INFRASTRUCTURE, see oracle/__init__.py).

torch-CPU restatement of the reference files

    mlx_vlm/models/llava_bunny/vision.py      SigLIP tower (Conv2d patch embed + learned positions, pre-LN encoder)
    mlx_vlm/models/llava_bunny/llava_bunny.py  mlp2x_gelu projector, <image> splice, get_input_embeddings
    mlx_vlm/models/llava_bunny/language.py    Qwen1.5 decoder (q/k/v bias, nn.RoPE, SwiGLU), lm_head
    mlx_vlm/models/base.py:121-194            BaseImageProcessor (resize 384 bicubic, 1/255, (x - 0.5) / 0.5)

on the primitives of oracle/ops.py (same typed-graph rounding policy).  Weight names are the reference's module tree
after `Model.sanitize` (llava_bunny.py:180-222).  Pinned by tests/test_oracle_ref_golden_bunny.py against vectors
produced by the reference's own files executed over oracle/mlx_shim (tests/golden/make_golden_ref_bunny.py).

What the reference computes and this file deliberately does not: the SigLIP pooling head (vision.py:196-226).  Its
result is the first element of the tower's return tuple, which `get_input_embeddings` discards (`*_, hidden_state`,
llava_bunny.py:113-117) - it cannot influence any output of the path.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

from . import ops

F32 = torch.float32


@dataclass
class VisionCfg:
    num_hidden_layers: int = 27
    hidden_size: int = 1152
    intermediate_size: int = 4304
    num_attention_heads: int = 16
    image_size: int = 384
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-6

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid


@dataclass
class TextCfg:
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    intermediate_size: int = 2816
    num_attention_heads: int = 16
    num_key_value_heads: int = 16
    rms_norm_eps: float = 1e-6
    vocab_size: int = 151936
    rope_theta: float = 1000000.0
    attention_bias: bool = True
    tie_word_embeddings: bool = True


@dataclass
class Cfg:
    text: TextCfg = field(default_factory=TextCfg)
    vision: VisionCfg = field(default_factory=VisionCfg)
    image_token_index: int = -200


def tiny_cfg() -> Cfg:
    """Real head dims (72 vision, 64 text) and the real 27 x 27 patch grid (the model asserts 729 image tokens,
    llava_bunny.py:118) at toy widths / depths."""
    return Cfg(text=TextCfg(hidden_size=128, num_hidden_layers=2, intermediate_size=256, num_attention_heads=2,
                            num_key_value_heads=2, vocab_size=1024),
               vision=VisionCfg(num_hidden_layers=2, hidden_size=144, intermediate_size=288, num_attention_heads=2))


# weight scales at which the tiny model's greedy continuations are varied (not one repeated token): used by the golden
# generator and the parity tests
TEST_WEIGHT_SCALES = dict(std=0.12, embed_std=0.03)

V = "vision_tower.vision_tower.vision_model."
LM = "language_model.model."


def random_weights(cfg: Cfg, seed: int = 0, dtype=torch.bfloat16, std: float = 0.05, embed_std: float = 0.2,
                   with_pooling_head: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded weights under the reference's (sanitized) names.  `with_pooling_head` adds the parameters of the unused
    pooling head so that the reference's strict `load_weights` accepts the tree."""
    g = torch.Generator().manual_seed(seed)
    v, t = cfg.vision, cfg.text

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    def ln(prefix, dim):
        return {prefix + ".weight": (1 + 0.1 * torch.randn(dim, generator=g)).to(dtype), prefix + ".bias": rn(dim, s=0.1)}

    W: Dict[str, torch.Tensor] = {}
    E, I = v.hidden_size, v.intermediate_size
    W[V + "embeddings.patch_embedding.weight"] = rn(E, v.patch_size, v.patch_size, v.num_channels)   # (O, kH, kW, C)
    W[V + "embeddings.patch_embedding.bias"] = rn(E, s=0.1)
    W[V + "embeddings.position_embedding.weight"] = rn(v.num_patches, E, s=0.1)
    for i in range(v.num_hidden_layers):
        p = f"{V}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            W[p + f"self_attn.{n}.weight"] = rn(E, E)
            W[p + f"self_attn.{n}.bias"] = rn(E, s=0.1)
        W.update(ln(p + "layer_norm1", E))
        W.update(ln(p + "layer_norm2", E))
        W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"] = rn(I, E), rn(I, s=0.1)
        W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"] = rn(E, I), rn(E, s=0.1)
    W.update(ln(V + "post_layernorm", E))
    D = t.hidden_size
    W["mm_projector.linear_1.weight"], W["mm_projector.linear_1.bias"] = rn(D, E), rn(D, s=0.1)
    W["mm_projector.linear_2.weight"], W["mm_projector.linear_2.bias"] = rn(D, D), rn(D, s=0.1)
    hd = D // t.num_attention_heads
    W[LM + "embed_tokens.weight"] = rn(t.vocab_size, D, s=embed_std)
    for i in range(t.num_hidden_layers):
        p = f"{LM}layers.{i}."
        for n, rows in (("q_proj", t.num_attention_heads * hd), ("k_proj", t.num_key_value_heads * hd),
                        ("v_proj", t.num_key_value_heads * hd)):
            W[p + f"self_attn.{n}.weight"] = rn(rows, D)
            if t.attention_bias:
                W[p + f"self_attn.{n}.bias"] = rn(rows, s=0.1)
        W[p + "self_attn.o_proj.weight"] = rn(D, t.num_attention_heads * hd)
        W[p + "mlp.gate_proj.weight"] = rn(t.intermediate_size, D)
        W[p + "mlp.up_proj.weight"] = rn(t.intermediate_size, D)
        W[p + "mlp.down_proj.weight"] = rn(D, t.intermediate_size)
        W[p + "input_layernorm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype)
        W[p + "post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype)
    W[LM + "norm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype)
    if not t.tie_word_embeddings:
        W[LM + "lm_head.weight"] = rn(t.vocab_size, D)
    if with_pooling_head:          # own generator: the weights above are the same with and without the head
        g = torch.Generator().manual_seed(seed + 1)
        W[V + "head.probe"] = rn(1, 1, E)
        W[V + "head.attention.in_proj.weight"], W[V + "head.attention.in_proj.bias"] = rn(3 * E, E), rn(3 * E, s=0.1)
        W[V + "head.attention.out_proj.weight"], W[V + "head.attention.out_proj.bias"] = rn(E, E), rn(E, s=0.1)
        W.update(ln(V + "head.layernorm", E))
        W[V + "head.mlp.fc1.weight"], W[V + "head.mlp.fc1.bias"] = rn(I, E), rn(I, s=0.1)
        W[V + "head.mlp.fc2.weight"], W[V + "head.mlp.fc2.bias"] = rn(E, I), rn(E, s=0.1)
    return W


# --------------------------------------------------------------------------------------------- image processor
def preprocess(images_hwc_u8, size: int = 384) -> np.ndarray:
    """`ImageProcessor.preprocess` (llava_bunny.py:24-57 over base.py:121-194): RGB -> resize to size x size (PIL
    bicubic, what transformers' `resize` does for uint8 input) -> x / 255 -> (x - 0.5) / 0.5 -> channels first.
    -> float32 [B, 3, size, size]."""
    from PIL import Image

    out = []
    for img in images_hwc_u8:
        pil = Image.fromarray(np.asarray(img, dtype=np.uint8)).convert("RGB").resize((size, size), resample=Image.BICUBIC)
        x = np.asarray(pil).astype(np.float64) * (1 / 255)          # transformers.rescale: float64 product, cast to f32
        x = x.astype(np.float32)
        x = (x - np.float32(0.5)) / np.float32(0.5)
        out.append(np.transpose(x, (2, 0, 1)))
    return np.stack(out).astype(np.float32)


# --------------------------------------------------------------------------------------------- SigLIP tower
def vision_embeddings(W, cfg: Cfg, pixel_values: torch.Tensor) -> torch.Tensor:
    """VisionEmbeddings (vision.py:147-173): Conv2d(k = s = patch, bias) over NHWC == one GEMM per patch with the
    patch flattened (kH, kW, C)-major, then `+= position_embedding` (a typed add).  pixel_values [B, 3, H, W]."""
    v = cfg.vision
    B = pixel_values.shape[0]
    P, G = v.patch_size, v.grid
    w = W[V + "embeddings.patch_embedding.weight"]
    x = pixel_values.permute(0, 2, 3, 1)[:, : G * P, : G * P]                    # NHWC; the conv drops the remainder
    x = x.reshape(B, G, P, G, P, v.num_channels).permute(0, 1, 3, 2, 4, 5).reshape(B, G * G, P * P * v.num_channels)
    y = ops.linear(x, w.reshape(w.shape[0], -1), W[V + "embeddings.patch_embedding.bias"])
    return ops.add(y, W[V + "embeddings.position_embedding.weight"][None])


def encoder_layer(W, i: int, cfg: Cfg, x: torch.Tensor) -> torch.Tensor:
    """EncoderLayer (vision.py:122-139) with Attention (27-78): pre-LN, full (unmasked) attention over the 729
    patches, scale = head_dim ** -0.5, FastGELUMLP (mlp.py:47-57)."""
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


def vision_tower(W, cfg: Cfg, pixel_values: torch.Tensor, return_layers: bool = False, embeddings=None):
    """SigLipVisionModel (vision.py:176-201) up to the last encoder state, which is what the model uses
    (`hidden_state[-1]`, llava_bunny.py:113-117).  -> [B, 729, E].  `embeddings`: start from given patch + position
    embeddings (tests: the reference's own, so that the encoder is compared bit for bit although Conv2d and the GEMM
    restatement of it sum in different orders)."""
    x = vision_embeddings(W, cfg, pixel_values) if embeddings is None else embeddings
    states = [x]
    for i in range(cfg.vision.num_hidden_layers):
        x = encoder_layer(W, i, cfg, x)
        states.append(x)
    return (x, states) if return_layers else x


def mm_projector(W, x: torch.Tensor) -> torch.Tensor:
    """LlavaMultiModalProjector (llava_bunny.py:60-74): Linear -> nn.GELU() (erf) -> Linear."""
    h = ops.gelu_erf(ops.linear(x, W["mm_projector.linear_1.weight"], W["mm_projector.linear_1.bias"]))
    return ops.linear(h, W["mm_projector.linear_2.weight"], W["mm_projector.linear_2.bias"])


# --------------------------------------------------------------------------------------------- language model
def embed_tokens(W, input_ids) -> torch.Tensor:
    return W[LM + "embed_tokens.weight"][torch.as_tensor(np.asarray(input_ids), dtype=torch.long)]


def get_input_embeddings(W, cfg: Cfg, input_ids, pixel_values: Optional[torch.Tensor] = None,
                         vision_embeds=None, cast_pixels: bool = True) -> torch.Tensor:
    """Model.get_input_embeddings + _prepare_inputs_for_multimodal (llava_bunny.py:99-156): row b's FIRST <image>
    position (argmax of the match mask) is replaced by the 729 projected features of image b.  The embedding lookup
    runs on the ids as given, -200 included (nn.Embedding wraps negative indices; that row is then cut out).

    cast_pixels=False is the reference AS SHIPPED: this model never casts `pixel_values` to the weight dtype (Qwen2-VL
    does, qwen2_vl.py:44-45) and `prepare_inputs` hands over float32 (utils.py:2090-2091), so with bf16 weights MLX's
    type promotion turns every activation from the patch embedding on into float32 - tower, projector
    (`.astype(pixel_values.dtype)`, llava_bunny.py:117), the spliced prompt, the prefill, the prompt's KV cache and,
    through the float32 cache, the decode steps.  cast_pixels=True is the bf16 typed graph (pixels cast first)."""
    ids = np.asarray(input_ids)
    emb = embed_tokens(W, ids)
    if pixel_values is None:
        return emb
    pix = pixel_values.to(emb.dtype) if cast_pixels else pixel_values
    feats = mm_projector(W, vision_tower(W, cfg, pix, embeddings=vision_embeds).to(pix.dtype))
    emb = emb.to(torch.promote_types(emb.dtype, feats.dtype))
    rows = []
    for b in range(ids.shape[0]):
        pos = int(np.argmax(ids[b] == cfg.image_token_index))
        rows.append(torch.cat([emb[b:b + 1, :pos], feats[b:b + 1], emb[b:b + 1, pos + 1:]], dim=1))
    return torch.cat(rows, dim=0)


def attention(W, p: str, cfg: Cfg, x: torch.Tensor, cache: Optional[ops.KVCache]) -> torch.Tensor:
    """Attention (language.py:15-77): q/k/v with bias, nn.RoPE(head_dim, traditional=False, base=rope_theta) at
    offset = cache.offset (mx.fast.rope: fp32 angles and rotation, one rounding), KVCache, causal SDPA, o_proj."""
    t = cfg.text
    B, L, D = x.shape
    H, Hkv = t.num_attention_heads, t.num_key_value_heads
    hd = D // H

    def proj(n, heads):
        b = W.get(p + n + ".bias")
        return ops.linear(x, W[p + n + ".weight"], b).reshape(B, L, heads, hd).permute(0, 2, 1, 3)

    q, k, v = proj("q_proj", H), proj("k_proj", Hkv), proj("v_proj", Hkv)
    off = cache.offset if cache is not None else 0
    pos = torch.arange(off, off + L)[None].expand(B, L)
    inv = ops.mrope_inv_freq(hd, t.rope_theta)
    q = ops.mrope_apply(q, pos, inv, None, "fused")          # 2-D positions: plain rotate-half RoPE
    k = ops.mrope_apply(k, pos, inv, None, "fused")
    if cache is not None:
        k, v = cache.update_and_fetch(k, v)
    o = ops.sdpa(q, k, v, scale=hd ** -0.5, causal=L > 1, q_offset=k.shape[2] - L)
    return ops.linear(o.permute(0, 2, 1, 3).reshape(B, L, -1), W[p + "o_proj.weight"])


def decoder_layer(W, i: int, cfg: Cfg, x: torch.Tensor, cache) -> torch.Tensor:
    """TransformerBlock (language.py:80-106)."""
    p = f"{LM}layers.{i}."
    eps = cfg.text.rms_norm_eps
    h = ops.add(x, attention(W, p + "self_attn.", cfg, ops.rms_norm(x, W[p + "input_layernorm.weight"], eps), cache))
    hn = ops.rms_norm(h, W[p + "post_attention_layernorm.weight"], eps)
    act = ops.swiglu(ops.linear(hn, W[p + "mlp.gate_proj.weight"]), ops.linear(hn, W[p + "mlp.up_proj.weight"]))
    return ops.add(h, ops.linear(act, W[p + "mlp.down_proj.weight"]))


def language_model(W, cfg: Cfg, inputs_embeds: torch.Tensor, cache=None) -> torch.Tensor:
    """Qwen2Model (language.py:109-145): layers -> norm -> lm_head on EVERY position.  -> logits [B, L, V]"""
    h = inputs_embeds
    cache = cache or [None] * cfg.text.num_hidden_layers
    for i in range(cfg.text.num_hidden_layers):
        h = decoder_layer(W, i, cfg, h, cache[i])
    h = ops.rms_norm(h, W[LM + "norm.weight"], cfg.text.rms_norm_eps)
    head = W[LM + "embed_tokens.weight"] if cfg.text.tie_word_embeddings else W[LM + "lm_head.weight"]
    return ops.linear(h, head)


def generate_greedy(W, cfg: Cfg, input_ids, pixel_values: Optional[torch.Tensor] = None, max_tokens: int = 8,
                    return_logits: bool = False, vision_embeds=None, cast_pixels: bool = True):
    """generate_step, temperature 0 (generate/ar.py:151-515): embeddings -> whole-prompt prefill -> last-row logits ->
    logprobs -> argmax -> one-token decode steps through the KVCache."""
    ids = np.asarray(input_ids)
    assert ids.shape[0] == 1
    cache = [ops.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    logits = language_model(W, cfg, get_input_embeddings(W, cfg, ids, pixel_values, vision_embeds, cast_pixels), cache)[:, -1, :]
    toks, rows = [], []
    for n in range(max_tokens):
        y = int(ops.argmax_first(ops.logprobs_from_logits(logits))[0])
        toks.append(y)
        rows.append(logits[0].clone())
        if n == max_tokens - 1:
            break
        logits = language_model(W, cfg, embed_tokens(W, np.array([[y]])), cache)[:, -1, :]
    return (toks, torch.stack(rows)) if return_logits else toks


def decode_teacher_forced(W, cfg: Cfg, input_ids, pixel_values: Optional[torch.Tensor] = None, forced_tokens=(),
                          cast_pixels: bool = True) -> torch.Tensor:
    """Test construction (no reference counterpart): generate_step's device work (generate/ar.py:334-389) with the FED
    tokens prescribed.  -> logits [1 + len(forced_tokens), V]: row 0 = last prompt row of the whole-prompt prefill, row
    i = after feeding forced_tokens[i - 1] through the KVCache.  The head is applied to the last row only (the
    reference computes every row and slices, ar.py:358 - same values, 700 x less work at V = 151,936)."""
    ids = np.asarray(input_ids)
    assert ids.shape[0] == 1
    t = cfg.text
    cache = [ops.KVCache() for _ in range(t.num_hidden_layers)]
    head = W[LM + "embed_tokens.weight"] if t.tie_word_embeddings else W[LM + "lm_head.weight"]

    def last_row_logits(h):
        for i in range(t.num_hidden_layers):
            h = decoder_layer(W, i, cfg, h, cache[i])
        return ops.linear(ops.rms_norm(h[:, -1:, :], W[LM + "norm.weight"], t.rms_norm_eps), head)[0, 0]

    rows = [last_row_logits(get_input_embeddings(W, cfg, ids, pixel_values, None, cast_pixels))]
    for y in forced_tokens:
        rows.append(last_row_logits(embed_tokens(W, np.array([[int(y)]]))))
    return torch.stack(rows)
