"""Qwen1.5 language model of nanoLLaVA on the decode / prefill engine - host mirror of the reference's
`mlx_vlm/models/llava_bunny/language.py` (Attention 15-77: q/k/v bias, nn.RoPE rotate-half at the cache offset;
TransformerBlock 80-106; Qwen2Model 109-145; sanitize 163-174).

The engine (`csrc/engine.hip`: fused RMSNorm+QKV+RoPE+KV-write GEMV, MFMA paged decode attention, ...) is built for
128-wide heads and rotates with per-frequency positions (M-RoPE).  This model has 64-wide heads and plain RoPE; both
map onto the engine without new kernels:

  * heads are laid out in 128 columns at load time: real dims [0, 32) -> columns [0, 32), real dims [32, 64) ->
    columns [64, 96), everything else zero (rows of q/k/v projections and biases; columns of o_proj).  The engine's
    rotate-half pairs column d with d + 64, i.e. exactly real dim d with d + 32; zero columns stay zero under rotation
    and add exact zeros to q.k and P.V.  `attn_scale = 64 ** -0.5` replaces the engine's 128 ** -0.5.
  * the frequency table holds theta ** (-2 i / 64) for the 32 real pairs and 0 for the rest (`rope_dim = 64`);
  * M-RoPE with the same position on all three axes is plain RoPE, so positions are arange(L) and rope_deltas 0.

Cost: K/V pages store 128 columns per head for 64 of content.  A 64-wide decode-attention kernel removes that; the
model is launch-bound at this size (0.5 B parameters), so it is not the first thing to fix.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import numpy as np
import torch

from ..qwen2_vl.language import LanguageModel as _Engine
from .config import ModelConfig, TextConfig

ENGINE_HEAD_DIM = 128


class LanguageModel(_Engine):
    ROTATING_POS_FROM_RING = False     # rope offset of a decode step over a rotating cache = cache.offset (the reference's own read)
    def __init__(self, args: TextConfig, config: ModelConfig, device="cuda", **engine_kwargs):
        hd = args.hidden_size // args.num_attention_heads
        if hd > ENGINE_HEAD_DIM or hd % 2:
            raise NotImplementedError(f"head_dim {hd}")
        if args.rope_traditional:
            raise NotImplementedError("rope_traditional (interleaved pairs) is outside the built path")
        rope_scale = 1.0
        if args.rope_scaling is not None and args.rope_scaling.get("type") == "linear":
            rope_scale = 1.0 / float(args.rope_scaling["factor"])
        if rope_scale != 1.0:
            raise NotImplementedError("linear rope scaling is outside the built path")
        self.text_config = args
        self.real_head_dim = hd
        eng = SimpleNamespace(model_type=args.model_type, hidden_size=args.hidden_size,
                              num_hidden_layers=args.num_hidden_layers, intermediate_size=args.intermediate_size,
                              num_attention_heads=args.num_attention_heads, num_key_value_heads=args.num_key_value_heads,
                              rms_norm_eps=args.rms_norm_eps, vocab_size=args.vocab_size, rope_theta=args.rope_theta,
                              rope_scaling=None, tie_word_embeddings=args.tie_word_embeddings,
                              head_dim=ENGINE_HEAD_DIM if hd != ENGINE_HEAD_DIM else None,
                              rope_dim=hd, attn_scale=float(hd) ** -0.5)
        super().__init__(eng, config, device=device, **engine_kwargs)

    # ------------------------------------------------------------------ weights
    def _spread(self, w: torch.Tensor, heads: int) -> torch.Tensor:
        """[heads * hd, ...] -> [heads * 128, ...]: the two rotary halves of each head go to columns [0, hd/2) and
        [64, 64 + hd/2) of its 128-wide slot."""
        hd, half = self.real_head_dim, self.real_head_dim // 2
        if hd == ENGINE_HEAD_DIM:
            return w
        tail = w.shape[1:]
        src = w.reshape(heads, hd, *tail)
        out = torch.zeros(heads, ENGINE_HEAD_DIM, *tail, dtype=w.dtype, device=w.device)
        out[:, :half] = src[:, :half]
        out[:, ENGINE_HEAD_DIM // 2: ENGINE_HEAD_DIM // 2 + half] = src[:, half:]
        return out.reshape(heads * ENGINE_HEAD_DIM, *tail)

    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: names relative to `language_model.` (`model.layers.i...`, `model.embed_tokens.weight`,
        `model.lm_head.weight` - the reference keeps lm_head inside `model`, language.py:121)."""
        t = self.text_config
        H, Hkv, hd = t.num_attention_heads, t.num_key_value_heads, self.real_head_dim
        out: Dict[str, torch.Tensor] = {}
        for k, v in W.items():
            if k.endswith(("self_attn.q_proj.weight", "self_attn.q_proj.bias")):
                out[k] = self._spread(v, H)
            elif k.endswith(("self_attn.k_proj.weight", "self_attn.k_proj.bias", "self_attn.v_proj.weight",
                             "self_attn.v_proj.bias")):
                out[k] = self._spread(v, Hkv)
            elif k.endswith("self_attn.o_proj.weight"):
                out[k] = self._spread(v.t(), H).t().contiguous()
            elif k == "model.lm_head.weight":
                out["lm_head.weight"] = v
            else:
                out[k] = v
        for i in range(t.num_hidden_layers):          # attention_bias=False checkpoints: zero biases
            for n, heads in (("q_proj", H), ("k_proj", Hkv), ("v_proj", Hkv)):
                key = f"model.layers.{i}.self_attn.{n}.bias"
                if key not in out:
                    out[key] = torch.zeros(heads * ENGINE_HEAD_DIM, dtype=torch.bfloat16)
        return super().load_weights(out)

    # ------------------------------------------------------------------ positions: plain RoPE
    def get_rope_index(self, input_ids, image_grid_thw=None, video_grid_thw=None, attention_mask=None):
        ids = np.asarray(input_ids)
        B, L = ids.shape
        pos = np.broadcast_to(np.arange(L, dtype=np.int64)[None, None], (3, B, L)).copy()
        return pos, np.zeros((B, 1), dtype=np.int64)

    def sanitize(self, weights):
        """reference language.py:163-174"""
        if self.text_config.tie_word_embeddings and "language_model.model.lm_head.weight" not in weights:
            weights["language_model.model.lm_head.weight"] = weights["language_model.model.embed_tokens.weight"]
        return {k: v for k, v in weights.items() if "self_attn.rotary_emb.inv_freq" not in k}
