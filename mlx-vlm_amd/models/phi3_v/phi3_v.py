"""Phi-3.5-vision glue model - host mirror of the reference's `mlx_vlm/models/phi3_v/phi3_v.py` (Model 174-250:
get_input_embeddings 199-233 - image positions are the NEGATIVE input ids, image i's projected rows are written from
the (sum of earlier counts)-th negative position on; `language_model` is the model itself in the reference, here the
engine-backed decoder that the generate loop drives) and of `vision.py`'s write-back (257-262).

Checkpoint names are the reference's module tree, which is also the HF layout of Phi-3.5-vision-instruct: `model.
embed_tokens`, `model.layers.N.*`, `model.norm`, `model.vision_embed_tokens.*`, `lm_head` - no renaming, only the conv
layout fix of `VisionModel.sanitize` (vision.py:264-280)."""
from __future__ import annotations

from typing import Dict, Iterable, List

import numpy as np
import torch

from ..base import InputEmbeddingsFeatures
from ..qwen2_vl.language import _to_np
from .config import ModelConfig
from .language import LanguageModel
from .vision import VisionModel

_VT = "model.vision_embed_tokens."


def sanitize_keys(keys: Iterable[str]) -> List[str]:
    """The names a checkpoint's keys end up with after `VisionModel.sanitize` (the model itself renames nothing)."""
    return [k for k in keys if "position_ids" not in k]


class Model:
    def __init__(self, config: ModelConfig, device="cuda", **engine_kwargs):
        self.config = config
        self.model_type = config.model_type
        self.device = device
        self.vision_model = VisionModel(config, device=device)
        self.vision_tower = self.vision_model            # load_model's generic sanitize hook looks for `vision_tower`
        self.language_model = LanguageModel(config, device=device, **engine_kwargs)

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights: Dict[str, torch.Tensor], strict: bool = True):
        vt = {k[len(_VT):]: v for k, v in weights.items() if k.startswith(_VT)}
        lm = {k: v for k, v in weights.items() if not k.startswith(_VT)}
        if strict:
            extra = [k for k in lm if not k.startswith(("model.embed_tokens.", "model.layers.", "model.norm.", "lm_head."))]
            if extra:
                raise ValueError(f"unexpected weight names: {extra[:5]}")
        self.vision_model.load_weights(vt)
        self.language_model.load_weights(lm)
        return self

    def eval(self):
        return self

    @property
    def layers(self):
        return self.language_model.layers

    @property
    def head_dim(self):
        return self.config.hidden_size // self.config.num_attention_heads

    @property
    def n_kv_heads(self):
        return self.config.num_key_value_heads

    def encode_image(self, pixel_values, image_sizes=None, **kwargs) -> torch.Tensor:
        """-> the projected rows of every image, concatenated (what a vision-feature cache stores for this model)"""
        return torch.cat(self.vision_model.image_features(pixel_values, image_sizes), dim=0)

    def encode_images_batched(self, pixel_values_list, extras):
        """The views of several requests through ONE CLIP pass (every view is 336 x 336) and one projection GEMM pair.
        -> per request: projected rows of its images, concatenated, or None"""
        idx = [j for j, pv in enumerate(pixel_values_list) if pv is not None]
        out = [None] * len(pixel_values_list)
        if not idx:
            return out
        pvs = [torch.as_tensor(pixel_values_list[j]) for j in idx]
        sizes = [np.asarray((extras[j] or {})["image_sizes"]).reshape(-1, 2) for j in idx]
        T = max(p.shape[1] for p in pvs)
        dev_any = any(p.is_cuda for p in pvs)
        pvs = [p if p.shape[1] == T else torch.cat([p, p.new_zeros(p.shape[0], T - p.shape[1], *p.shape[2:])], dim=1) for p in pvs]
        if dev_any:
            pvs = [p if p.is_cuda else p.to(self.device) for p in pvs]
        rows = self.vision_model.image_features(torch.cat(pvs, dim=0), np.concatenate(sizes, axis=0))
        at = 0
        for j, sz in zip(idx, sizes):
            out[j] = torch.cat(rows[at: at + len(sz)], dim=0) if len(sz) > 1 else rows[at]
            at += len(sz)
        return out

    # ------------------------------------------------------------------ reference phi3_v.py:199-233
    def get_input_embeddings(self, input_ids=None, pixel_values=None, **kwargs):
        lm = self.language_model
        ids = _to_np(input_ids)
        if ids.ndim == 1:
            ids = ids[None]
        emb = lm.embed_tokens(np.where(ids < 0, 0, ids))               # the rows at negative ids are overwritten below
        pos, deltas = lm.get_rope_index(ids)
        if pixel_values is not None:
            where = np.argwhere(ids < 0)
            cached = kwargs.get("cached_image_features", None)
            if cached is not None:
                rows = cached
            else:
                image_sizes = kwargs.get("image_sizes", None)
                if image_sizes is None:
                    raise ValueError("phi3_v needs `image_sizes` next to `pixel_values` (the processor returns both)")
                rows = self.vision_model.image_features(pixel_values, image_sizes)
                rows = torch.cat(rows, dim=0) if len(rows) > 1 else rows[0]
            # image i's rows go to its run of negative ids (the processor emits exactly cnt_i of them, contiguous): the
            # reference writes cnt_i rows from the run's first position on (vision.py:257-262) - the same rows
            n = min(rows.shape[0], len(where))
            from ... import _lib
            flat = _lib.h2d((where[:n, 0] * ids.shape[1] + where[:n, 1]).astype(np.int64), self.device)
            emb.view(-1, emb.shape[-1]).index_copy_(0, flat, rows[:n].to(emb.dtype))
        return InputEmbeddingsFeatures(inputs_embeds=emb, position_ids=pos, rope_deltas=deltas)

    def __call__(self, input_ids, pixel_values=None, mask=None, cache=None, **kwargs):
        f = self.get_input_embeddings(input_ids, pixel_values, **kwargs)
        return self.language_model(input_ids, inputs_embeds=f.inputs_embeds, cache=cache, mask=None,
                                   position_ids=f.position_ids)

    # ------------------------------------------------------------------ checkpoint names
    def sanitize(self, weights):
        return self.vision_model.sanitize(weights)
