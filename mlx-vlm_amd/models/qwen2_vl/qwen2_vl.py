"""Qwen2-VL glue model on MI355X - host mirror of the reference's
mlx_vlm/models/qwen2_vl/qwen2_vl.py (Model: get_input_embeddings,
merge_input_ids_with_image_features, sanitize, __call__)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ... import _lib, ops
from ..base import InputEmbeddingsFeatures
from .config import ModelConfig
from .language import LanguageModel, _to_np
from .vision import VisionModel


class Model:
    def __init__(self, config: ModelConfig, device="cuda", **engine_kwargs):
        self.config = config
        self.device = device
        self.vision_tower = VisionModel(config.vision_config, device=device)
        self.language_model = LanguageModel(config.text_config, config, device=device, **engine_kwargs)

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights: Dict[str, torch.Tensor], strict: bool = True):
        """weights: sanitized names (`vision_tower.*`, `language_model.*`), any device/dtype."""
        vt = {k[len("vision_tower."):]: v for k, v in weights.items() if k.startswith("vision_tower.")}
        lm = {k[len("language_model."):]: v for k, v in weights.items() if k.startswith("language_model.")}
        if strict and (len(vt) + len(lm) != len(weights)):
            extra = [k for k in weights if not k.startswith(("vision_tower.", "language_model."))]
            raise ValueError(f"unexpected weight names: {extra[:5]}")
        # a 4-bit checkpoint may quantize the tower's Linears too (utils.py:961 predicate): the tower runs on the bf16 MFMA
        # GEMMs, so those weights are materialised once, on the device (vlm_dequant_w4)
        from .. import quantized as Qz
        for path in [k[:-len(".scales")] for k in list(vt) if k.endswith(".scales")]:
            vt[path + ".weight"] = Qz.dequantize_bf16(Qz.take(vt, path), self.device)
            del vt[path + ".scales"], vt[path + ".biases"]
        self.vision_tower.load_weights(vt)
        self.language_model.load_weights(lm)
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ reference qwen2_vl.py:20-76
    def get_input_embeddings(self, input_ids=None, pixel_values=None, **kwargs):
        if pixel_values is None:
            pixel_values = kwargs.get("pixel_values_videos", None)
        image_grid_thw = kwargs.get("image_grid_thw", None)
        video_grid_thw = kwargs.get("video_grid_thw", None)
        mask = kwargs.get("mask", None)
        grid_thw = image_grid_thw if image_grid_thw is not None else video_grid_thw
        ids = _to_np(input_ids)
        if pixel_values is None:
            position_ids, rope_deltas = self.language_model.get_rope_index(ids, attention_mask=mask)
            return InputEmbeddingsFeatures(inputs_embeds=self.language_model.embed_tokens(ids),
                                           position_ids=position_ids, rope_deltas=rope_deltas)
        inputs_embeds = self.language_model.embed_tokens(ids)
        cached = kwargs.get("cached_image_features", None)
        if cached is not None:
            hidden_states = cached
        else:
            hidden_states = self.vision_tower(torch.as_tensor(pixel_values), _to_np(grid_thw), output_hidden_states=False)
        final = self.merge_input_ids_with_image_features(self.config.image_token_id, self.config.video_token_id,
                                                         hidden_states, inputs_embeds, ids)
        position_ids, rope_deltas = self.language_model.get_rope_index(ids, image_grid_thw, video_grid_thw, mask)
        return InputEmbeddingsFeatures(inputs_embeds=final, position_ids=position_ids, rope_deltas=rope_deltas)

    def encode_image(self, pixel_values, image_grid_thw=None, **kwargs):
        """Projected image features (vision tower + PatchMerger) - what `cached_image_features` carries (reference hook
        `model.encode_image`, dispatch.py:805-809)."""
        return self.vision_tower(torch.as_tensor(pixel_values), _to_np(image_grid_thw), output_hidden_states=False)

    @staticmethod
    def merge_input_ids_with_image_features(image_token_id, video_token_id, image_features, inputs_embeds, input_ids):
        """reference qwen2_vl.py:78-148: row-major over the batch, the i-th image-token position receives image
        feature row i.  Positions are found on the host (ids are host resident); the copy is one scatter kernel."""
        ids = _to_np(input_ids)
        pos = ids == image_token_id
        if pos.sum() == 0:
            pos = ids == video_token_id
        n = int(pos.sum())
        if n == 0:
            return inputs_embeds
        if image_features.shape[0] != n:
            raise ValueError(
                f"Number of image token positions ({n}) does not match number of image features ({image_features.shape[0]})")
        B, Lq, D = inputs_embeds.shape
        rows = np.nonzero(pos.reshape(-1))[0].astype(np.int32)
        rows_d = _lib.h2d(rows, inputs_embeds.device)
        flat = inputs_embeds.reshape(B * Lq, D)
        ops.scatter_rows_(image_features.contiguous(), rows_d, flat)
        return flat.view(B, Lq, D)

    @property
    def layers(self):
        return self.language_model.layers

    def __call__(self, input_ids, pixel_values=None, mask=None, cache=None, **kwargs):
        f = self.get_input_embeddings(input_ids, pixel_values, **kwargs)
        kwargs = {"pixel_values": pixel_values, **kwargs}
        return self.language_model(input_ids, f.inputs_embeds, mask=mask, cache=cache, **kwargs)

    def sanitize(self, weights):
        """reference qwen2_vl.py:179-190 (+ the transformers>=4.5x `model.visual.` / `model.language_model.` layout)."""
        def transform_key(key):
            if key.startswith("model.visual."):
                key = "visual." + key[len("model.visual."):]
            elif key.startswith("model.language_model."):
                key = "model." + key[len("model.language_model."):]
            if "vision_tower" not in key:
                key = key.replace("visual", "vision_tower")
            if "language_model" not in key:
                if "model" in key:
                    key = key.replace("model", "language_model.model")
                elif "lm_head" in key:
                    key = key.replace("lm_head", "language_model.lm_head")
            return key

        return {transform_key(k): v for k, v in weights.items()}
