"""Shared model dataclasses - same fields and behaviour as the reference's
mlx_vlm/models/base.py:54-118 (LanguageModelOutput, InputEmbeddingsFeatures,
BaseModelConfig), holding torch device tensors instead of mx.arrays."""
from __future__ import annotations

import inspect
from dataclasses import dataclass
from typing import Any, Dict, List, Optional


@dataclass
class LanguageModelOutput:
    logits: Any
    hidden_states: Optional[List[Any]] = None
    cross_attention_states: Optional[List[Any]] = None
    encoder_outputs: Optional[List[Any]] = None
    gdn_states: Optional[List] = None
    shared_kv_states: Optional[Dict[str, tuple]] = None


@dataclass
class InputEmbeddingsFeatures:
    inputs_embeds: Any
    attention_mask_4d: Optional[Any] = None
    visual_pos_masks: Optional[Any] = None
    deepstack_visual_embeds: Optional[Any] = None
    per_layer_inputs: Optional[Any] = None
    cross_attention_states: Optional[Any] = None
    cross_attention_mask: Optional[Any] = None
    full_text_row_masked_out_mask: Optional[Any] = None
    decoder_inputs_embeds: Optional[Any] = None
    attention_mask: Optional[Any] = None
    position_ids: Optional[Any] = None
    pos_hw: Optional[Any] = None
    rope_deltas: Optional[Any] = None

    def to_dict(self):
        return {k: getattr(self, k) for k in self.__dataclass_fields__}


@dataclass
class BaseModelConfig:
    @classmethod
    def from_dict(cls, params):
        if not params:
            return cls()
        return cls(**{k: v for k, v in params.items() if k in inspect.signature(cls).parameters})

    def to_dict(self):
        return {k: v for k, v in self.__dict__.items() if v is not None}
