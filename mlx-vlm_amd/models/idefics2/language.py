"""Mistral decoder of Idefics2 on the decode / prefill engine - host mirror of the reference's
`mlx_vlm/models/idefics2/language.py` (Attention 15-70: bias-free q / k / v, nn.RoPE rotate-half at the cache offset, GQA;
TransformerBlock 73-98; LanguageModel 101-141: untied lm_head; sanitize 143-147).

Mistral-7B has the engine's native 128-wide heads; everything else is what models/llava_bunny/language.py already maps
onto the engine (plain RoPE as equal-axis M-RoPE, zero q / k / v biases, narrower heads spread into 128 columns), so this
class only translates the reference's parameter names (`layers.N...`, `embed_tokens`, `norm`, `lm_head` directly under
`language_model.`) into the ones that loader takes."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch

from ..llava_bunny.language import LanguageModel as _PlainRopeEngine
from .config import TextConfig


class LanguageModel(_PlainRopeEngine):
    def __init__(self, args: TextConfig, config=None, device="cuda", **engine_kwargs):
        t = SimpleNamespace(model_type=args.model_type, hidden_size=args.hidden_size, num_hidden_layers=args.num_hidden_layers,
                            intermediate_size=args.intermediate_size, num_attention_heads=args.num_attention_heads,
                            num_key_value_heads=args.num_key_value_heads, rms_norm_eps=args.rms_norm_eps,
                            vocab_size=args.vocab_size, rope_theta=args.rope_theta, rope_traditional=args.rope_traditional,
                            rope_scaling=None, attention_bias=False, tie_word_embeddings=bool(args.tie_word_embeddings))
        super().__init__(t, config, device=device, **engine_kwargs)
        self.config = args

    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: names relative to `language_model.` as the reference's module tree has them"""
        out = {}
        for k, v in W.items():
            if k.startswith("lm_head."):
                out["model." + k] = v                       # the loader below keeps the head inside `model`
            elif k.startswith(("layers.", "embed_tokens.", "norm.")):
                out["model." + k] = v
            else:
                out[k] = v
        return super().load_weights(out)

    def sanitize(self, weights):
        """reference language.py:143-147"""
        return {k: v for k, v in weights.items() if "self_attn.rotary_emb.inv_freq" not in k}
