"""Qwen2-VL configuration objects for the drop-in module contract.

The reference's `config.json` files (HF layout: text parameters at the root, `vision_config` nested) must load
unchanged, so the three classes expose the same field NAMES and DEFAULTS as the reference's
`mlx_vlm/models/qwen2_vl/config.py:8-77` (that is the contract: `ModelConfig.from_dict(config.json)`,
`.text_config`, `.vision_config`).  They are generated from the field tables below instead of being spelled out
as dataclass bodies; validation and the root-level -> text_config lifting live in plain functions.
"""
from __future__ import annotations

from dataclasses import field, make_dataclass
from typing import Any, Dict, List, Optional, Tuple

from ..base import BaseModelConfig

_REQUIRED = object()

# (name, type, default) - order matters: positional construction mirrors the reference's classes
_VISION: Tuple[Tuple[str, Any, Any], ...] = (
    ("model_type", str, "qwen2_vl"), ("depth", int, 32), ("embed_dim", int, 1280), ("hidden_size", int, 1536),
    ("num_heads", int, 16), ("image_size", int, 384), ("patch_size", int, 14), ("vocab_size", int, 32000),
    ("mlp_ratio", float, 4.0), ("in_channels", int, 3), ("layer_norm_eps", float, 1e-6),
    ("spatial_patch_size", int, 14), ("spatial_merge_size", int, 2), ("temporal_patch_size", int, 2),
)
_TEXT: Tuple[Tuple[str, Any, Any], ...] = (
    ("model_type", str, _REQUIRED), ("hidden_size", int, _REQUIRED), ("num_hidden_layers", int, _REQUIRED),
    ("intermediate_size", int, _REQUIRED), ("num_attention_heads", int, _REQUIRED), ("rms_norm_eps", float, _REQUIRED),
    ("vocab_size", int, _REQUIRED), ("num_key_value_heads", Optional[int], 8),
    ("max_position_embeddings", Optional[int], 40960), ("rope_theta", float, 1000000.0),
    ("rope_traditional", bool, False), ("rope_scaling", Optional[Dict[str, Any]], None),
    ("tie_word_embeddings", bool, False), ("sliding_window", int, 32768), ("use_sliding_window", bool, False),
    ("use_cache", bool, True),
)
_MODEL: Tuple[Tuple[str, Any, Any], ...] = (
    ("text_config", Any, _REQUIRED), ("vision_config", Any, _REQUIRED), ("model_type", str, _REQUIRED),
    ("ignore_index", int, -100), ("image_token_id", int, 151655), ("video_token_id", int, 151656),
    ("vision_start_token_id", int, 151652), ("vision_feature_select_strategy", str, "default"),
    ("vision_feature_layer", int, -2), ("vocab_size", int, 32000), ("eos_token_id", Optional[List[int]], None),
)


def _build(name: str, table, namespace=None):
    specs = [(n, t) if d is _REQUIRED else (n, t, field(default=d)) for n, t, d in table]
    cls = make_dataclass(name, specs, bases=(BaseModelConfig,), namespace=namespace or {})
    cls.__module__ = __name__
    return cls


def _check_text(self):
    """GQA default and the M-RoPE description every Qwen2-VL checkpoint carries."""
    if self.num_key_value_heads is None:
        self.num_key_value_heads = self.num_attention_heads
    rs = self.rope_scaling
    if rs:
        missing = {"mrope_section", "type"} - set(rs)
        if missing:
            raise ValueError(f"rope_scaling must contain keys {{'mrope_section', 'type'}} (missing {sorted(missing)})")
        if rs["type"] not in ("mrope", "default"):
            raise ValueError("rope_scaling type must be 'mrope' or 'default'")


VisionConfig = _build("VisionConfig", _VISION)
TextConfig = _build("TextConfig", _TEXT, {"__post_init__": _check_text})


def _model_from_dict(cls, params):
    """HF Qwen2-VL `config.json`: every root-level key except `vision_config` describes the language model."""
    known = {n for n, _, _ in _MODEL}
    raw = dict(params)
    text = raw.get("text_config")
    if not text or not isinstance(text, (dict, TextConfig)):
        text = {k: v for k, v in raw.items() if k != "vision_config"}
    raw["text_config"] = text
    cfg = cls(**{k: v for k, v in raw.items() if k in known})
    if isinstance(cfg.text_config, dict):
        cfg.text_config = TextConfig.from_dict(cfg.text_config)
    if isinstance(cfg.vision_config, dict):
        cfg.vision_config = VisionConfig.from_dict(cfg.vision_config)
    return cfg


ModelConfig = _build("ModelConfig", _MODEL, {"from_dict": classmethod(_model_from_dict)})

__all__ = ["ModelConfig", "TextConfig", "VisionConfig"]
