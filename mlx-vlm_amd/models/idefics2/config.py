"""Idefics2 configuration objects: same field names, defaults and `from_dict` behaviour as the reference's
`mlx_vlm/models/idefics2/config.py:7-65` (three nested configs - text = Mistral, vision = SigLIP-style tower, perceiver -
and `image_token_index` defaulting to `image_token_id`)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

from ..base import BaseModelConfig


@dataclass
class VisionConfig(BaseModelConfig):
    model_type: str = "idefics2"
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    num_channels: int = 3
    image_size: int = 224
    patch_size: int = 32
    layer_norm_eps: float = 1e-6


@dataclass
class TextConfig(BaseModelConfig):
    model_type: str = "mistral"
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: Optional[int] = 8
    rms_norm_eps: float = 1e-5
    vocab_size: int = 32003
    rope_theta: float = 1000000.0
    rope_traditional: bool = False
    max_position_embeddings: int = 32768
    tie_word_embeddings: bool = False

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads


@dataclass
class PerceiverConfig(BaseModelConfig):
    model_type: str = "idefics2"
    num_key_value_heads: int = 4
    resampler_depth: int = 3
    resampler_head_dim: int = 96
    resampler_n_heads: int = 16
    resampler_n_latents: int = 64


@dataclass
class ModelConfig(BaseModelConfig):
    text_config: TextConfig = None
    vision_config: VisionConfig = None
    perceiver_config: PerceiverConfig = None
    model_type: str = "idefics2"
    ignore_index: int = -100
    image_token_id: int = 32001
    vocab_size: int = 151936
    image_token_index: Optional[int] = None
    eos_token_id: Optional[List[int]] = None
    quantization: Optional[dict] = None

    def __post_init__(self):
        if self.image_token_index is None:
            self.image_token_index = self.image_token_id
        if isinstance(self.text_config, dict) or self.text_config is None:
            self.text_config = TextConfig.from_dict(self.text_config or {})
        if isinstance(self.vision_config, dict) or self.vision_config is None:
            self.vision_config = VisionConfig.from_dict(self.vision_config or {})
        if isinstance(self.perceiver_config, dict) or self.perceiver_config is None:
            self.perceiver_config = PerceiverConfig.from_dict(self.perceiver_config or {})


__all__ = ["ModelConfig", "TextConfig", "VisionConfig", "PerceiverConfig"]
