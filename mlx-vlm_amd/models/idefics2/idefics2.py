"""Idefics2 glue model - host mirror of the reference's `mlx_vlm/models/idefics2/idefics2.py` (Model 170-321):
get_input_embeddings (padding-image removal: an all-zero image is padding; patch mask from the pixel mask; tower;
connector; `masked_scatter` of the resampler rows into the `<image>` positions, 15-33 + 264-284) and sanitize (294-321).

Numerics note (the llava_bunny one again): the reference never casts `pixel_values` to the weight dtype here either, so the
float32 pixels its own pipeline produces promote the tower and the connector to float32 activations (bf16 weights), rounded
once when the features are scattered into the bf16 prompt.  This engine computes the bf16 typed graph (pixels cast first);
`oracle/idefics2.py` restates both (`cast_pixels`), tests/test_oracle_ref_golden_idefics2.py pins both to the reference."""
from __future__ import annotations

import re
from typing import Dict, Iterable, List

import numpy as np
import torch

from ... import _lib
from ..base import InputEmbeddingsFeatures
from ..qwen2_vl.language import _to_np
from .config import ModelConfig
from .connector import Connector
from .language import LanguageModel
from .vision import VisionModel


def _rename(k: str) -> str:
    if k.startswith("model."):
        k = k.split(".", 1)[1]
    elif k.startswith("lm_head."):
        k = "language_model." + k
    if k.startswith("text_model."):
        k = "language_model." + k.split(".", 1)[1]
    return k


def sanitize_keys(keys: Iterable[str]) -> List[str]:
    """The names a checkpoint's keys end up with after Model.sanitize + LanguageModel.sanitize (+ VisionModel.sanitize, which
    renames nothing).  Pure name logic."""
    return [k for k in (_rename(k) for k in keys) if "self_attn.rotary_emb.inv_freq" not in k]


class Model:
    def __init__(self, config: ModelConfig, device="cuda", **engine_kwargs):
        self.config = config
        self.model_type = config.model_type
        self.device = device
        self.vision_model = VisionModel(config.vision_config, device=device)
        self.vision_tower = self.vision_model                 # load_model's generic sanitize hook
        self.language_model = LanguageModel(config.text_config, config, device=device, **engine_kwargs)
        self.connector = Connector(config, device=device)

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights: Dict[str, torch.Tensor], strict: bool = True):
        groups = {"vision_model.": {}, "connector.": {}, "language_model.": {}}
        for k, v in weights.items():
            for pre, d in groups.items():
                if k.startswith(pre):
                    d[k[len(pre):]] = v
                    break
            else:
                if strict:
                    raise ValueError(f"unexpected weight name: {k}")
        self.vision_model.load_weights(groups["vision_model."])
        self.connector.load_weights(groups["connector."])
        self.language_model.load_weights(groups["language_model."])
        return self

    def eval(self):
        return self

    @property
    def layers(self):
        return self.language_model.layers

    # ------------------------------------------------------------------ reference idefics2.py:184-250
    def _real_images(self, pixel_values, pixel_attention_mask):
        """-> (real images [n, C, H, W] float32 - a device tensor when the pixels already live there, else a host array -,
        patch mask bool [n, gh, gw])"""
        if isinstance(pixel_values, torch.Tensor) and pixel_values.is_cuda:
            B, N, C, H, W = pixel_values.shape
            pv = pixel_values.reshape(B * N, C, H, W)
            real = np.where((pv != 0).reshape(B * N, -1).any(dim=1).cpu().numpy())[0]      # bookkeeping: B * N flags come back
            if len(real) != B * N:
                pv = pv[torch.as_tensor(real, device=pv.device)]
        else:
            pv = pixel_values.detach().float().numpy() if isinstance(pixel_values, torch.Tensor) else np.asarray(pixel_values, dtype=np.float32)
            B, N, C, H, W = pv.shape
            pv = pv.reshape(B * N, C, H, W)
            real = np.where((pv == 0.0).reshape(B * N, -1).sum(axis=1) != C * H * W)[0]
            pv = pv[real]
        if pixel_attention_mask is None:
            pam = np.ones((len(real), H, W), dtype=bool)
        else:
            pam = _to_np(pixel_attention_mask).reshape(B * N, H, W)[real] > 0
        P = self.config.vision_config.patch_size
        gh, gw = H // P, W // P
        r = pam[:, : gh * P, : gw * P].reshape(len(real), gh, P, gw, P)
        return pv, r.transpose(0, 1, 3, 2, 4).sum(axis=(-1, -2)) > 0

    def encode_image(self, pixel_values, pixel_attention_mask=None, **kwargs) -> torch.Tensor:
        """-> resampler outputs of every real image, bf16 [n * n_latents, hidden]"""
        pv, pmask = self._real_images(pixel_values, pixel_attention_mask)
        pooled = self.vision_model(pv if isinstance(pv, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pv)), pmask)
        return self.connector(pooled, pv.shape[0])

    def encode_images_batched(self, pixel_values_list, extras):
        """The images of several requests through ONE tower + connector pass per padded image size (requests whose images
        were padded to different sizes cannot share a pass: padding patches take part in the unmasked encoder attention,
        as in the reference).  -> per request: resampler rows [n_i * n_latents, hidden] or None"""
        groups, out = {}, [None] * len(pixel_values_list)
        for j, (pv, kw) in enumerate(zip(pixel_values_list, extras)):
            if pv is None:
                continue
            real, pmask = self._real_images(pv, (kw or {}).get("pixel_attention_mask", None))
            groups.setdefault(tuple(real.shape[1:]), []).append((j, real, pmask))
        nl = self.config.perceiver_config.resampler_n_latents
        for items in groups.values():
            reals = [r if isinstance(r, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(r)) for _, r, _ in items]
            if any(r.is_cuda for r in reals):
                reals = [r if r.is_cuda else r.to(self.device) for r in reals]
            real = torch.cat(reals, dim=0) if len(reals) > 1 else reals[0]
            pmask = np.concatenate([m for _, _, m in items], axis=0)
            feats = self.connector(self.vision_model(real, pmask), real.shape[0])
            at = 0
            for j, r, _ in items:
                out[j] = feats[at * nl: (at + r.shape[0]) * nl]
                at += r.shape[0]
        return out

    def get_input_embeddings(self, input_ids=None, pixel_values=None, **kwargs):
        lm = self.language_model
        ids = _to_np(input_ids)
        if ids.ndim == 1:
            ids = ids[None]
        emb = lm.embed_tokens(ids)
        pos, deltas = lm.get_rope_index(ids)
        if pixel_values is not None:
            cached = kwargs.get("cached_image_features", None)
            feats = cached if cached is not None else self.encode_image(pixel_values, kwargs.get("pixel_attention_mask", None))
            where = np.argwhere(ids == self.config.image_token_index)
            if len(where) != feats.shape[0]:
                raise ValueError(f"Image features and image tokens do not match: tokens: {len(where)}, features {feats.shape[0]}")
            flat_rows = _lib.h2d((where[:, 0] * ids.shape[1] + where[:, 1]).astype(np.int64), self.device)
            emb.view(-1, emb.shape[-1]).index_copy_(0, flat_rows, feats.to(emb.dtype))          # masked_scatter: row copies
        return InputEmbeddingsFeatures(inputs_embeds=emb, position_ids=pos, rope_deltas=deltas)

    def __call__(self, input_ids, pixel_values=None, mask=None, cache=None, **kwargs):
        f = self.get_input_embeddings(input_ids, pixel_values, **kwargs)
        return self.language_model(input_ids, inputs_embeds=f.inputs_embeds, cache=cache, mask=None, position_ids=f.position_ids)

    # ------------------------------------------------------------------ checkpoint names (reference 294-321)
    def sanitize(self, weights):
        out = {_rename(k): v for k, v in weights.items()}
        out = self.language_model.sanitize(out)
        return {k: (self.vision_model.sanitize({k: v})[k] if k.startswith("vision_model.") else v) for k, v in out.items()}
