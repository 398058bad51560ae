"""nanoLLaVA glue model - host mirror of the reference's `mlx_vlm/models/llava_bunny/llava_bunny.py`
(ImageProcessor 24-57, LlavaMultiModalProjector 60-74, Model.get_input_embeddings 99-127,
_prepare_inputs_for_multimodal 129-156, sanitize 180-222).

Numerics note.  The reference never casts `pixel_values` to the weight dtype in this model (it does in Qwen2-VL,
qwen2_vl.py:44-45), and its input pipeline hands over float32 (utils.py:2090-2091): with bf16 weights MLX's type
promotion then carries float32 activations through the tower, the projector, the prefill and - through the float32
prompt KV cache - the decode steps.  This engine computes the bf16 typed graph (pixels cast to bf16 first), i.e. what
the reference computes when it is handed bf16 pixels.  `oracle/llava_bunny.py` restates both; the tests state the
tolerance against each."""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from ... import _lib, ops
from ..base import InputEmbeddingsFeatures
from ..qwen2_vl.language import _to_np
from .config import ModelConfig
from .language import LanguageModel
from .vision import VisionModel

_V = "vision_tower.vision_tower.vision_model."

# (pattern on the HF key, replacement) - first match wins; the rest keep their name
_KEY_RULES = (
    (re.compile(r"^model\.(vision_tower.*)$"), r"\1"),                         # drop the leading `model.`
    (re.compile(r"^model\.mm_projector\.0\.(weight|bias)$"), r"mm_projector.linear_1.\1"),
    (re.compile(r"^model\.mm_projector\.2\.(weight|bias)$"), r"mm_projector.linear_2.\1"),
    (re.compile(r"^(lm_head.*)$"), r"language_model.model.\1"),
    (re.compile(r"^(model\.(?:embed_tokens|norm|layers).*)$"), r"language_model.\1"),
)
_HEAD_RULES = (
    (re.compile(r"^(" + re.escape(_V) + r"head\.attention\.in_proj)_bias.*$"), r"\1.bias"),
    (re.compile(r"^(" + re.escape(_V) + r"head\.attention\.in_proj)_weight.*$"), r"\1.weight"),
)


def _rename(key: str) -> str:
    for pat, rep in _KEY_RULES:
        if pat.match(key):
            key = pat.sub(rep, key)
            break
    for pat, rep in _HEAD_RULES:
        if pat.match(key):
            return pat.sub(rep, key)
    return key


def sanitize_keys(keys: Iterable[str], tie_word_embeddings: bool = True) -> List[str]:
    """The names a checkpoint's keys end up with after Model.sanitize + LanguageModel.sanitize + VisionModel.sanitize
    (what `load_weights` is then given).  Pure name logic, usable without a device."""
    out = [_rename(k) for k in keys]
    if tie_word_embeddings and "language_model.model.lm_head.weight" not in out \
            and "language_model.model.embed_tokens.weight" in out:
        out.append("language_model.model.lm_head.weight")
    return [k for k in out if "self_attn.rotary_emb.inv_freq" not in k and "position_ids" not in k]


class Model:
    def __init__(self, config: ModelConfig, device="cuda", **engine_kwargs):
        self.config = config
        self.model_type = config.model_type
        self.device = device
        self.vision_tower = VisionModel(config.vision_config, device=device)
        self.language_model = LanguageModel(config.text_config, config, device=device, **engine_kwargs)
        self._proj: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ weights
    def load_weights(self, weights: Dict[str, torch.Tensor], strict: bool = True):
        """weights under the sanitized names: `vision_tower.vision_tower.vision_model.*`, `mm_projector.*`,
        `language_model.*`.  The pooling head of the tower (`...vision_model.head.*`, `post_layernorm`) is accepted and
        not used: its output is discarded by the model (llava_bunny.py:113-117)."""
        vt = {k[len(_V):]: v for k, v in weights.items() if k.startswith(_V)}
        lm = {k[len("language_model."):]: v for k, v in weights.items() if k.startswith("language_model.")}
        pj = {k[len("mm_projector."):]: v for k, v in weights.items() if k.startswith("mm_projector.")}
        if strict and len(vt) + len(lm) + len(pj) != len(weights):
            extra = [k for k in weights if not k.startswith((_V, "language_model.", "mm_projector."))]
            raise ValueError(f"unexpected weight names: {extra[:5]}")
        self.vision_tower.load_weights(vt)
        self.language_model.load_weights(lm)
        self._proj = {k: v.to(device=self.device, dtype=torch.bfloat16).contiguous() for k, v in pj.items()}
        return self

    def eval(self):
        return self

    @property
    def layers(self):
        return self.language_model.layers

    # ------------------------------------------------------------------ projector (reference llava_bunny.py:60-74)
    def mm_projector(self, x: torch.Tensor) -> torch.Tensor:
        p = self._proj
        h = ops.gemm(x, p["linear_1.weight"], bias=p["linear_1.bias"], epilogue=ops.EPI_BIAS | ops.EPI_GELU_ERF)
        return ops.gemm(h, p["linear_2.weight"], bias=p["linear_2.bias"], epilogue=ops.EPI_BIAS)

    def encode_image(self, pixel_values) -> torch.Tensor:
        """-> projected image features bf16 [B * 729, hidden]"""
        return self.mm_projector(self.vision_tower(torch.as_tensor(pixel_values)))

    # ------------------------------------------------------------------ reference llava_bunny.py:99-156
    def get_input_embeddings(self, input_ids=None, pixel_values=None, **kwargs):
        lm = self.language_model
        ids = _to_np(input_ids)
        if pixel_values is None:
            pos, deltas = lm.get_rope_index(ids)
            return InputEmbeddingsFeatures(inputs_embeds=lm.embed_tokens(np.where(ids < 0, 0, ids)), position_ids=pos,
                                           rope_deltas=deltas)
        cached = kwargs.get("cached_image_features", None)
        feats = cached if cached is not None else self.encode_image(pixel_values)
        Np = self.vision_tower.num_patches
        B, L = ids.shape
        if feats.shape[0] != B * Np:
            raise ValueError(f"{feats.shape[0]} image feature rows for {B} prompts x {Np} patches")
        # the FIRST <image> position of each row (argmax of the match mask; 0 when there is none, as the reference)
        where = [int(np.argmax(ids[b] == self.config.image_token_index)) for b in range(B)]
        emb = lm.embed_tokens(np.where(ids < 0, 0, ids))                       # the sentinel's row is cut out below
        D = emb.shape[-1]
        out = torch.empty(B, L - 1 + Np, D, dtype=emb.dtype, device=emb.device)
        for b, pos in enumerate(where):                                          # three row-range copies per prompt
            out[b, :pos].copy_(emb[b, :pos])
            out[b, pos:pos + Np].copy_(feats[b * Np:(b + 1) * Np])
            out[b, pos + Np:].copy_(emb[b, pos + 1:])
        full = np.zeros((B, L - 1 + Np), dtype=np.int64)
        position_ids, deltas = lm.get_rope_index(full)
        return InputEmbeddingsFeatures(inputs_embeds=out, position_ids=position_ids, rope_deltas=deltas)

    def __call__(self, input_ids, pixel_values=None, mask=None, cache=None, **kwargs):
        f = self.get_input_embeddings(input_ids, pixel_values)
        return self.language_model(input_ids, inputs_embeds=f.inputs_embeds, cache=cache, mask=None,
                                   position_ids=f.position_ids)

    # ------------------------------------------------------------------ checkpoint names (reference 180-222)
    def sanitize(self, weights):
        return {_rename(k): v for k, v in weights.items()}
