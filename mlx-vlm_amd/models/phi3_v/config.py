"""Phi-3.5-vision (`phi3_v`) configuration objects: same field names, defaults and `from_dict` behaviour as the
reference's `mlx_vlm/models/phi3_v/config.py:6-83` - the text parameters sit at the ROOT of the HF config.json (this
model's `text_config` carries only `max_position_embeddings`), the vision tower's dimensions are literals of the model
(CLIP ViT-L/14-336, vision.py:182-192), and the chat EOS ids 2 / 32000 / 32007 are appended to `eos_token_id` whenever
the vocabulary holds them (config.py:11-27)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

from ..base import BaseModelConfig

PHI3_V_CHAT_EOS_TOKEN_IDS = [2, 32000, 32007]


def _normalize_eos(eos_token_id, vocab_size):
    if isinstance(eos_token_id, int):
        ids = [eos_token_id]
    elif eos_token_id is None:
        ids = []
    else:
        ids = list(eos_token_id)
    if vocab_size > max(PHI3_V_CHAT_EOS_TOKEN_IDS):
        ids += [t for t in PHI3_V_CHAT_EOS_TOKEN_IDS if t not in ids]
    return ids or eos_token_id


@dataclass
class TextConfig(BaseModelConfig):
    max_position_embeddings: int = 4096


@dataclass
class VisionConfig(BaseModelConfig):
    model_type: str = "phi3_v"
    num_hidden_layers: int = 24
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_attention_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    projection_dim: int = 768
    vocab_size: int = 32000
    num_channels: int = 3
    layer_norm_eps: float = 1e-5
    image_dim_out: int = 1024
    model_name: str = "openai/clip-vit-large-patch14-336"
    name: str = "clip_vision_model"
    num_img_tokens: int = 144


@dataclass
class ModelConfig(BaseModelConfig):
    text_config: TextConfig = field(default_factory=TextConfig)
    vision_config: VisionConfig = field(default_factory=VisionConfig)
    model_type: str = "phi3_v"
    vocab_size: int = 32064
    num_hidden_layers: int = 32
    intermediate_size: int = 8192
    num_attention_heads: int = 32
    rms_norm_eps: float = 1e-5
    ignore_index: int = -100
    image_token_index: int = 257152
    hidden_size: int = 2048
    pad_token_id: int = 0
    num_key_value_heads: Optional[int] = None
    rope_theta: float = 10000
    rope_traditional: bool = False
    partial_rotary_factor: float = 1.0
    rope_scaling: Optional[Dict[str, Union[float, str, List[float]]]] = None
    max_position_embeddings: int = 131072
    original_max_position_embeddings: int = 4096
    eos_token_id: Optional[Union[int, List[int]]] = None
    tie_word_embeddings: bool = False
    quantization: Optional[dict] = None

    def __post_init__(self):
        self.eos_token_id = _normalize_eos(self.eos_token_id, self.vocab_size or 0)
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        if isinstance(self.text_config, dict):
            self.text_config = TextConfig.from_dict(self.text_config)
        if isinstance(self.vision_config, dict):
            self.vision_config = VisionConfig.from_dict(self.vision_config)


__all__ = ["ModelConfig", "TextConfig", "VisionConfig", "PHI3_V_CHAT_EOS_TOKEN_IDS"]
