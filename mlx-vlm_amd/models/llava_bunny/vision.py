"""SigLIP vision tower of nanoLLaVA on the C-ABI kernels - host mirror of the reference's
`mlx_vlm/models/llava_bunny/vision.py` (VisionEmbeddings 147-173, EncoderLayer 122-139, Attention 27-78,
FastGELUMLP `models/mlp.py:47-57`), up to the last encoder state, which is all the model uses
(`hidden_state[-1]`, llava_bunny.py:113-117; the pooling head's output is discarded there and is not computed).

Per layer: LayerNorm -> one fused q|k|v GEMM (+bias) -> flash attention over each image's 729 patches ->
out_proj GEMM (+bias +residual) -> LayerNorm -> fc1 GEMM (+bias +GELU fast) -> fc2 GEMM (+bias +residual).

Head width: SigLIP-so400m heads are 72 wide; the attention kernel has 64 / 80 / 128.  Every head is zero-padded to 80
at load time (8 zero rows per head in the q / k / v projections and their biases, 8 zero columns per head in
out_proj): the padded coordinates contribute exact zeros to q.k and to P.V, the softmax scale stays 72 ** -0.5."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ... import _lib, ops
from .config import VisionConfig

_KERNEL_HEAD_DIMS = (64, 80, 128)


class VisionModel:
    def __init__(self, config: VisionConfig, device="cuda"):
        if config.model_type != "siglip_vision_model":
            raise ValueError(f"Unsupported model type: {config.model_type}")
        self.config = config
        self.model_type = config.model_type
        self.device = device
        c = config
        self.grid = c.image_size // c.patch_size
        self.num_patches = self.grid * self.grid
        self.head_dim = c.hidden_size // c.num_attention_heads
        self.head_pad = next(d for d in _KERNEL_HEAD_DIMS if d >= self.head_dim)
        self.patch_dim = c.patch_size * c.patch_size * c.num_channels
        self.patch_k = (self.patch_dim + 63) // 64 * 64
        self._w: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ weights
    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: names relative to `vision_tower.vision_tower.vision_model.` (patch weight (O, kH, kW, C), as the
        reference's sanitize leaves it)."""
        self._enc = None            # (the native layer loop's weight table is rebuilt on first use)
        c, dev, bf = self.config, self.device, torch.bfloat16
        E, H, hd, hp = c.hidden_size, c.num_attention_heads, self.head_dim, self.head_pad

        def g(name):
            return W[name].to(device=dev, dtype=bf)

        wp = torch.zeros(E, self.patch_k, dtype=bf, device=dev)
        wp[:, : self.patch_dim] = g("embeddings.patch_embedding.weight").reshape(E, -1)
        self._w.update(wpatch=wp, bpatch=g("embeddings.patch_embedding.bias").contiguous(),
                       pos=g("embeddings.position_embedding.weight").contiguous())

        def pad_rows(w):          # [H * hd, K] -> [H * hp, K]
            out = torch.zeros(H, hp, w.shape[1], dtype=bf, device=dev)
            out[:, :hd] = w.reshape(H, hd, -1)
            return out.reshape(H * hp, -1)

        def pad_vec(b):
            out = torch.zeros(H, hp, dtype=bf, device=dev)
            out[:, :hd] = b.reshape(H, hd)
            return out.reshape(-1)

        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}."
            wqkv = torch.cat([pad_rows(g(p + f"self_attn.{n}.weight")) for n in ("q_proj", "k_proj", "v_proj")], dim=0)
            bqkv = torch.cat([pad_vec(g(p + f"self_attn.{n}.bias")) for n in ("q_proj", "k_proj", "v_proj")], dim=0)
            wo = torch.zeros(E, H, hp, dtype=bf, device=dev)
            wo[:, :, :hd] = g(p + "self_attn.out_proj.weight").reshape(E, H, hd)
            self._w.update({f"{i}.wqkv": wqkv.contiguous(), f"{i}.bqkv": bqkv.contiguous(),
                            f"{i}.wo": wo.reshape(E, H * hp).contiguous(), f"{i}.bo": g(p + "self_attn.out_proj.bias"),
                            f"{i}.ln1w": g(p + "layer_norm1.weight"), f"{i}.ln1b": g(p + "layer_norm1.bias"),
                            f"{i}.ln2w": g(p + "layer_norm2.weight"), f"{i}.ln2b": g(p + "layer_norm2.bias"),
                            f"{i}.w1": g(p + "mlp.fc1.weight").contiguous(), f"{i}.b1": g(p + "mlp.fc1.bias"),
                            f"{i}.w2": g(p + "mlp.fc2.weight").contiguous(), f"{i}.b2": g(p + "mlp.fc2.bias")})
        return self

    # ------------------------------------------------------------------ forward
    def patchify(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[B, 3, H, W] float -> bf16 [B * 729, patch_k]: each row one patch flattened (kH, kW, C)-major, which is what
        the NHWC Conv2d with kernel = stride = patch contracts (the remainder of 384 / 14 is dropped, as the conv does)."""
        c, G, P = self.config, self.grid, self.config.patch_size
        x = pixel_values
        if not x.is_cuda:
            x = _lib.h2d(x, self.device)
        B = x.shape[0]
        x = x[:, :, : G * P, : G * P].to(torch.float32).reshape(B, c.num_channels, G, P, G, P)
        x = x.permute(0, 2, 4, 3, 5, 1).reshape(B * G * G, self.patch_dim).contiguous()      # data movement only
        return ops.cast_pad(x, self.patch_k)

    def __call__(self, pixel_values: torch.Tensor, output_hidden_states: Optional[bool] = None) -> torch.Tensor:
        """pixel_values [B, 3, 384, 384] (channels first, as the processor emits them).  -> last encoder state as
        bf16 [B * 729, hidden]."""
        c, w = self.config, self._w
        E, H, hp, Np = c.hidden_size, c.num_attention_heads, self.head_pad, self.num_patches
        patches = self.patchify(pixel_values)
        B = patches.shape[0] // Np
        x = torch.empty(B * Np, E, dtype=torch.bfloat16, device=self.device)
        for b in range(B):            # + bias, then + position table: the table is the residual operand of each image
            ops.gemm(patches[b * Np:(b + 1) * Np], w["wpatch"], bias=w["bpatch"], res=w["pos"], out=x[b * Np:(b + 1) * Np],
                     epilogue=ops.EPI_BIAS | ops.EPI_RESIDUAL)
        cu = _lib.h2d(np.arange(B + 1, dtype=np.int32) * Np, self.device)
        nqb = B * ((Np + 127) // 128)
        scale = float(self.head_dim) ** -0.5
        # the encoder layers as ONE native call (vlm_encoder_forward: 7 launches per layer, no host work in between)
        if getattr(self, "_enc", None) is None:
            self._enc = ops.EncoderLayers(w, c.num_hidden_layers)
        return self._enc.forward_(x, H, hp, c.layer_norm_eps, ops.EPI_GELU_FAST, cu, nqb, scale)

    # ------------------------------------------------------------------ checkpoint fix-ups (reference vision.py:243-266)
    def sanitize(self, weights):
        out = {}
        for k, v in weights.items():
            if "position_ids" in k:
                continue
            if "patch_embedding.weight" in k and v.ndim == 4:
                O, a, b_, c_ = v.shape
                if not (O >= a and O >= b_ and a == b_):       # torch layout (O, C, kH, kW) -> (O, kH, kW, C)
                    v = v.permute(0, 2, 3, 1)
            out[k] = v
        return out
