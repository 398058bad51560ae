"""nanoLLaVA (`llava_bunny`) configuration objects: same field names, defaults and `from_dict` behaviour as the
reference's `mlx_vlm/models/llava_bunny/config.py:9-85` (HF config.json: text parameters at the root, `vision_config`
nested, vision model_type defaulting to `siglip_vision_model`).  Generated from field tables like models/qwen2_vl."""
from __future__ import annotations

from dataclasses import field, make_dataclass
from typing import Any, Dict, List, Optional, Union

from ..base import BaseModelConfig

_REQ = object()

_TEXT = (
    ("model_type", str, _REQ), ("hidden_size", int, _REQ), ("num_hidden_layers", int, _REQ),
    ("intermediate_size", int, _REQ), ("num_attention_heads", int, _REQ), ("rms_norm_eps", float, _REQ),
    ("vocab_size", int, _REQ), ("attention_bias", bool, True), ("num_key_value_heads", Optional[int], None),
    ("rope_theta", float, 1000000.0), ("rope_traditional", bool, False),
    ("rope_scaling", Optional[Dict[str, Union[float, str]]], None), ("max_position_embeddings", int, 4096),
    ("tie_word_embeddings", bool, True),
)
_VISION = (
    ("model_type", str, _REQ), ("num_hidden_layers", int, 27), ("hidden_size", int, 1152),
    ("intermediate_size", int, 4304), ("num_attention_heads", int, 16), ("image_size", int, 384),
    ("patch_size", int, 14), ("projection_dim", int, 768), ("vocab_size", int, 32000), ("num_channels", int, 3),
    ("layer_norm_eps", float, 1e-6),
)
_MODEL = (
    ("text_config", Any, _REQ), ("vision_config", Any, _REQ), ("model_type", str, _REQ), ("auto_map", dict, _REQ),
    ("hidden_size", int, _REQ), ("mm_hidden_size", int, _REQ), ("mm_projector_type", str, "mlp2x_gelu"),
    ("ignore_index", int, -100), ("image_token_index", int, -200), ("vocab_size", int, 151936),
    ("eos_token_id", Optional[List[int]], None),
)


def _build(name, table, namespace=None):
    specs = [(n, t) if d is _REQ else (n, t, field(default=d)) for n, t, d in table]
    cls = make_dataclass(name, specs, bases=(BaseModelConfig,), namespace=namespace or {})
    cls.__module__ = __name__
    return cls


def _check_text(self):
    if self.num_key_value_heads is None:
        self.num_key_value_heads = self.num_attention_heads
    if self.rope_scaling:
        if not {"factor", "type"} <= set(self.rope_scaling):
            raise ValueError("rope_scaling must contain keys {'factor', 'type'}")
        if self.rope_scaling["type"] != "linear":
            raise ValueError("rope_scaling 'type' currently only supports 'linear'")


TextConfig = _build("TextConfig", _TEXT, {"__post_init__": _check_text})
VisionConfig = _build("VisionConfig", _VISION)


def _model_from_dict(cls, params):
    raw = dict(params)
    if not raw.get("text_config"):
        raw["text_config"] = {k: v for k, v in raw.items() if k != "vision_config"}
    vision = dict(raw.get("vision_config") or {})
    vision.setdefault("model_type", "siglip_vision_model")
    if not vision["model_type"]:
        vision["model_type"] = "siglip_vision_model"
    raw["vision_config"] = vision
    known = {n for n, _, _ in _MODEL}
    cfg = cls(**{k: v for k, v in raw.items() if k in known})
    if isinstance(cfg.text_config, dict):
        cfg.text_config = TextConfig.from_dict(cfg.text_config)
    if isinstance(cfg.vision_config, dict):
        cfg.vision_config = VisionConfig.from_dict(cfg.vision_config)
    return cfg


ModelConfig = _build("ModelConfig", _MODEL, {"from_dict": classmethod(_model_from_dict)})

__all__ = ["ModelConfig", "TextConfig", "VisionConfig"]
