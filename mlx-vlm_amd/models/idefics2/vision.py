"""Vision tower of Idefics2 on the C-ABI kernels - host mirror of the reference's `mlx_vlm/models/idefics2/vision.py`
(VisionEmbeddings 114-172: Conv2d patch embed + position ids bucketed from the patch mask; EncoderLayer 84-99 / Attention
27-81 / FastGELUMLP; VisionModel 175-205: `post_layernorm` of the last state with MLX's default eps).

All real images of a call share one padded H x W (the image processor pads to the largest), so every image is one segment
of gh x gw tokens.  The reference hands NO mask to the encoder (vision.py:196-199: the patch mask only shapes the position
ids), so attention runs over all patches of an image, padding patches included - as here.  Position ids follow the
reference's arithmetic, negative buckets included (they index the table from its end, oracle/idefics2.py::position_ids).

Head width: SigLIP-so400m heads are 72 wide, zero-padded to the 80 the attention kernel has (as models/llava_bunny)."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from ... import _lib, ops
from .config import VisionConfig

_KERNEL_HEAD_DIMS = (64, 80, 128)


def bucket_position_ids(patch_mask: np.ndarray, side: int) -> np.ndarray:
    """vision.py:143-166 -> int64 [n, gh * gw] (may be negative: an index from the end of the table)"""
    n, gh, gw = patch_mask.shape
    bounds = np.linspace(1 / side, 1.0, side, endpoint=False)
    ids = np.zeros((n, gh * gw), dtype=np.int64)
    for b in range(n):
        m = patch_mask[b]
        nh, nw = int(m[:, 0].sum()), int(m[0, :].sum())
        bh = np.digitize(np.linspace(0, 1, nh, endpoint=False), bounds, right=True) - 1
        bw = np.digitize(np.linspace(0, 1, nw, endpoint=False), bounds, right=True) - 1
        ids[b][m.reshape(-1)] = (bh[:, None] * side + bw).flatten()
    return ids


class VisionModel:
    def __init__(self, config: VisionConfig, device="cuda"):
        if config.model_type not in ("idefics2", "idefics2_vision"):
            raise ValueError(f"Unsupported model type: {config.model_type}")
        self.config = config
        self.model_type = config.model_type
        self.device = device
        c = config
        self.side = c.image_size // c.patch_size
        self.head_dim = c.hidden_size // c.num_attention_heads
        self.head_pad = next(d for d in _KERNEL_HEAD_DIMS if d >= self.head_dim)
        self.patch_dim = c.patch_size * c.patch_size * c.num_channels
        self.patch_k = (self.patch_dim + 63) // 64 * 64
        self._w: Dict[str, torch.Tensor] = {}

    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: names relative to `vision_model.`; patch weight (O, kH, kW, C) as `sanitize` leaves it."""
        self._enc = None            # (the native layer loop's weight table is rebuilt on first use)
        c, dev, bf = self.config, self.device, torch.bfloat16
        E, H, hd, hp = c.hidden_size, c.num_attention_heads, self.head_dim, self.head_pad

        from .. import quantized as Qz

        def g(name):
            if name.endswith(".weight") and Qz.has_scales(W, name[: -len(".weight")]):      # a 4-bit Linear: dequantised once
                return Qz.dequantize_bf16(Qz.take(W, name[: -len(".weight")]), dev)
            return W[name].to(device=dev, dtype=bf)

        wp = torch.zeros(E, self.patch_k, dtype=bf, device=dev)
        wp[:, : self.patch_dim] = g("embeddings.patch_embedding.weight").reshape(E, -1)
        self._w.update(wpatch=wp, bpatch=g("embeddings.patch_embedding.bias").contiguous(),
                       pos=g("embeddings.position_embedding.weight").contiguous(),
                       postw=g("post_layernorm.weight"), postb=g("post_layernorm.bias"))

        def pad_rows(w):
            out = torch.zeros(H, hp, w.shape[1], dtype=bf, device=dev)
            out[:, :hd] = w.reshape(H, hd, -1)
            return out.reshape(H * hp, -1)

        def pad_vec(b):
            out = torch.zeros(H, hp, dtype=bf, device=dev)
            out[:, :hd] = b.reshape(H, hd)
            return out.reshape(-1)

        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}."
            wo = torch.zeros(E, H, hp, dtype=bf, device=dev)
            wo[:, :, :hd] = g(p + "self_attn.out_proj.weight").reshape(E, H, hd)
            self._w.update({
                f"{i}.wqkv": torch.cat([pad_rows(g(p + f"self_attn.{n}.weight")) for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous(),
                f"{i}.bqkv": torch.cat([pad_vec(g(p + f"self_attn.{n}.bias")) for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous(),
                f"{i}.wo": wo.reshape(E, H * hp).contiguous(), f"{i}.bo": g(p + "self_attn.out_proj.bias"),
                f"{i}.ln1w": g(p + "layer_norm1.weight"), f"{i}.ln1b": g(p + "layer_norm1.bias"),
                f"{i}.ln2w": g(p + "layer_norm2.weight"), f"{i}.ln2b": g(p + "layer_norm2.bias"),
                f"{i}.w1": g(p + "mlp.fc1.weight").contiguous(), f"{i}.b1": g(p + "mlp.fc1.bias"),
                f"{i}.w2": g(p + "mlp.fc2.weight").contiguous(), f"{i}.b2": g(p + "mlp.fc2.bias")})
        return self

    def patchify(self, images: torch.Tensor) -> torch.Tensor:
        """[n, 3, H, W] float -> bf16 [n * gh * gw, patch_k]: a row = one patch flattened (kH, kW, C)-major"""
        c, P = self.config, self.config.patch_size
        x = images if images.is_cuda else _lib.h2d(images, self.device)
        n, _, H, W = x.shape
        gh, gw = H // P, W // P
        x = x[:, :, : gh * P, : gw * P].to(torch.float32).reshape(n, c.num_channels, gh, P, gw, P)
        x = x.permute(0, 2, 4, 3, 5, 1).reshape(n * gh * gw, self.patch_dim).contiguous()
        return ops.cast_pad(x, self.patch_k)

    def __call__(self, images: torch.Tensor, patch_attention_mask: np.ndarray) -> torch.Tensor:
        """images [n, 3, H, W] (channels first, the real images of a call), patch_attention_mask bool [n, gh, gw]
        -> pooler_output = post_layernorm(last state), bf16 [n * gh * gw, E]"""
        c, w = self.config, self._w
        E, H, hp = c.hidden_size, c.num_attention_heads, self.head_pad
        n = images.shape[0]
        L = patch_attention_mask.shape[1] * patch_attention_mask.shape[2]
        ids = bucket_position_ids(np.asarray(patch_attention_mask, dtype=bool), self.side) % w["pos"].shape[0]
        pos_rows = w["pos"].index_select(0, _lib.h2d(ids.reshape(-1).astype(np.int64), self.device))       # gather: data movement
        # conv rows + bias, then + the gathered position rows (a typed add: the residual operand)
        x = ops.gemm(self.patchify(images), w["wpatch"], bias=w["bpatch"], res=pos_rows, epilogue=ops.EPI_BIAS | ops.EPI_RESIDUAL)
        cu = _lib.h2d(np.arange(n + 1, dtype=np.int32) * L, self.device)
        nqb = n * ((L + 127) // 128)
        scale = float(self.head_dim) ** -0.5
        # the encoder layers as ONE native call (vlm_encoder_forward: LN, qkv GEMM, varlen flash attention, out GEMM +
        # residual, LN, fc1 + GELU, fc2 + residual per layer, no host work in between)
        if getattr(self, "_enc", None) is None:
            self._enc = ops.EncoderLayers(w, c.num_hidden_layers)
        self._enc.forward_(x, H, hp, c.layer_norm_eps, ops.EPI_GELU_FAST, cu, nqb, scale)
        return ops.layernorm(x, w["postw"], w["postb"], 1e-5)               # nn.LayerNorm(hidden): MLX's default eps

    def sanitize(self, weights):
        """reference vision.py:207-222: torch conv layout (O, C, kH, kW) -> (O, kH, kW, C)"""
        out = {}
        for k, v in weights.items():
            if "patch_embedding.weight" in k and v.ndim == 4:
                O, a, b_, _c = v.shape
                if not (O >= a and O >= b_ and a == b_):
                    v = v.permute(0, 2, 3, 1)
            out[k] = v
        return out
