"""Idefics2 configuration objects: same field names, defaults and `from_dict` behaviour as the reference's
`mlx_vlm/models/idefics2/config.py:7-65` (three nested configs - text = Mistral, vision = SigLIP-style tower, perceiver -
and `image_token_index` defaulting to `image_token_id`).  Generated from field tables like models/llava_bunny/config.py."""
from __future__ import annotations

from dataclasses import field, make_dataclass
from typing import Any, List, Optional

from ..base import BaseModelConfig

# (name, type, default) - the defaults of the reference's dataclasses (which are Mistral-7B-shaped even for the tower: a real
# config.json overrides them all)
_VISION = (("model_type", str, "idefics2"), ("hidden_size", int, 4096), ("intermediate_size", int, 14336),
           ("num_hidden_layers", int, 32), ("num_attention_heads", int, 32), ("num_key_value_heads", int, 8),
           ("num_channels", int, 3), ("image_size", int, 224), ("patch_size", int, 32), ("layer_norm_eps", float, 1e-6))
_TEXT = (("model_type", str, "mistral"), ("hidden_size", int, 4096), ("intermediate_size", int, 14336),
         ("num_hidden_layers", int, 32), ("num_attention_heads", int, 32), ("num_key_value_heads", Optional[int], 8),
         ("rms_norm_eps", float, 1e-5), ("vocab_size", int, 32003), ("rope_theta", float, 1000000.0),
         ("rope_traditional", bool, False), ("max_position_embeddings", int, 32768), ("tie_word_embeddings", bool, False))
_PERCEIVER = (("model_type", str, "idefics2"), ("num_key_value_heads", int, 4), ("resampler_depth", int, 3),
              ("resampler_head_dim", int, 96), ("resampler_n_heads", int, 16), ("resampler_n_latents", int, 64))
_MODEL = (("text_config", Any, None), ("vision_config", Any, None), ("perceiver_config", Any, None),
          ("model_type", str, "idefics2"), ("ignore_index", int, -100), ("image_token_id", int, 32001),
          ("vocab_size", int, 151936), ("image_token_index", Optional[int], None), ("eos_token_id", Optional[List[int]], None),
          ("quantization", Optional[dict], None))


def _build(name, table, post_init=None):
    ns = {"__post_init__": post_init} if post_init else {}
    cls = make_dataclass(name, [(n, t, field(default=d)) for n, t, d in table], bases=(BaseModelConfig,), namespace=ns)
    cls.__module__ = __name__
    return cls


def _text_post(self):
    if self.num_key_value_heads is None:
        self.num_key_value_heads = self.num_attention_heads


VisionConfig = _build("VisionConfig", _VISION)
TextConfig = _build("TextConfig", _TEXT, _text_post)
PerceiverConfig = _build("PerceiverConfig", _PERCEIVER)


def _model_post(self):
    if self.image_token_index is None:
        self.image_token_index = self.image_token_id
    for attr, cls in (("text_config", TextConfig), ("vision_config", VisionConfig), ("perceiver_config", PerceiverConfig)):
        v = getattr(self, attr)
        if v is None or isinstance(v, dict):
            setattr(self, attr, cls.from_dict(v or {}))


ModelConfig = _build("ModelConfig", _MODEL, _model_post)

__all__ = ["ModelConfig", "TextConfig", "VisionConfig", "PerceiverConfig"]
