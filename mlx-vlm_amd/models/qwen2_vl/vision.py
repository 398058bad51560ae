"""Qwen2-VL vision tower on MI355X - host mirror of the reference's
mlx_vlm/models/qwen2_vl/vision.py (VisionModel, same constructor / call /
sanitize contract).  The numerical work is one call into the native ViT engine
(csrc/engine.hip: patch GEMM -> 32 x [LN, qkv GEMM+bias, 2-D rope, varlen flash
attention, proj GEMM+bias+residual, LN, fc1 GEMM+bias+GELU, fc2 GEMM+bias+
residual] -> PatchMerger); the host only prepares integer position tables
(rot_pos_emb, cu_seqlens) from `grid_thw`, which the reference also does on the
host with .tolist() (vision.py:63,219-279).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from ... import _lib
from ..._lib import check
from .config import VisionConfig


def check_array_shape(arr) -> bool:
    """reference vision.py:9-26 (is the conv weight already channels-last?)."""
    shape = arr.shape
    if len(shape) not in [4, 5]:
        return False
    B, out_channels, kH, KW, t = shape
    if t == 3:
        return True
    return bool((out_channels >= kH) and (out_channels >= KW) and (kH == KW))


def rot_pos_ids(grid_thw: np.ndarray, merge: int) -> np.ndarray:
    """Integer part of VisionModel.rot_pos_emb (reference vision.py:219-247)."""
    out = []
    for t, h, w in np.asarray(grid_thw).tolist():
        hpos = np.repeat(np.arange(h)[:, None], w, axis=1)
        wpos = np.repeat(np.arange(w)[None, :], h, axis=0)

        def window(a):
            return a.reshape(h // merge, merge, w // merge, merge).transpose(0, 2, 1, 3).reshape(-1)

        out.append(np.tile(np.stack([window(hpos), window(wpos)], axis=-1), (t, 1)))
    return np.concatenate(out, axis=0).astype(np.int64)


class VisionModel:
    def __init__(self, config: VisionConfig, device="cuda"):
        self.config = config
        self.model_type = config.model_type
        if self.model_type != "qwen2_vl":
            raise ValueError(f"Unsupported model type: {self.model_type}")
        self.spatial_merge_size = config.spatial_merge_size
        self.device = device
        self.embed_dim = config.embed_dim
        self.num_heads = config.num_heads
        self.head_dim = config.embed_dim // config.num_heads
        self.patch_dim = config.in_channels * config.temporal_patch_size * config.patch_size * config.patch_size
        self.patch_k = (self.patch_dim + 63) // 64 * 64          # K of the patch GEMM, zero padded
        self.mlp_hidden = int(config.embed_dim * config.mlp_ratio)
        self._handle = None
        self._w: Dict[str, torch.Tensor] = {}
        self._tab_cache: Dict[tuple, tuple] = {}

    # ------------------------------------------------------------------ weights
    @property
    def dtype(self):
        return torch.bfloat16

    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: sanitized names relative to the tower (`patch_embed.proj.weight`, `blocks.i....`, `merger....`)."""
        c, dev = self.config, self.device
        L = _lib.lib()

        def g(name):
            t = W[name].to(device=dev, dtype=torch.bfloat16).contiguous()
            self._w[name] = t
            return t

        pw = W["patch_embed.proj.weight"]
        if pw.dim() == 5:
            # (O, T, H, W, C) channels-last as the reference holds it after sanitize.  The reference moves the
            # PIXELS to channels-last on every call (vision.py:93-101); we fold that permutation into the weight
            # once: columns back to the processor's (C, T, ph, pw) order, so pixel rows feed the GEMM as they are.
            pw = pw.permute(0, 4, 1, 2, 3).reshape(pw.shape[0], -1)
        wp = torch.zeros(c.embed_dim, self.patch_k, dtype=torch.bfloat16, device=dev)
        wp[:, : self.patch_dim] = pw.to(device=dev, dtype=torch.bfloat16)
        self._w["patch"] = wp

        # 2-D rope in the qkv GEMM epilogue (vlm_gemm_bf16_rope2d): the q / k rows of every block are interleaved per head
        # at load, (d, d + hd/2) -> (2d, 2d + 1).  q.k does not change under a common permutation of the head dimension,
        # v and everything downstream are untouched.  VLM_VIT_ROPE_FUSED=0: separate rope pass (A/B knob).
        import os
        self.rope_fused = os.environ.get("VLM_VIT_ROPE_FUSED", "1") != "0" and self.head_dim % 16 == 0
        cfg = _lib.VitConfig(c.depth, c.embed_dim, c.num_heads, self.mlp_hidden, self.patch_k, c.spatial_merge_size,
                             c.hidden_size, float(c.layer_norm_eps) if hasattr(c, "layer_norm_eps") else 1e-6,
                             int(self.rope_fused))
        cfg.ln_eps = 1e-6  # nn.LayerNorm(eps=1e-6) is hard-coded in the reference (vision.py:109,180-181)
        E, hd, half = c.embed_dim, self.head_dim, self.head_dim // 2
        inter = torch.stack([torch.arange(half), torch.arange(half) + half], dim=1).reshape(-1)          # 0, hd/2, 1, hd/2+1, ...
        perm = (torch.arange(2 * c.num_heads)[:, None] * hd + inter[None, :]).reshape(-1)                   # q and k heads
        self._qk_perm = torch.cat([perm, torch.arange(2 * E, 3 * E)])
        h = C.c_void_p()
        check(L.vlm_vit_create(C.byref(cfg), C.byref(h)), "vit_create")
        self._handle = h
        for i in range(c.depth):
            p = f"blocks.{i}."
            if self.rope_fused:
                W = dict(W)
                for n in ("attn.qkv.weight", "attn.qkv.bias"):
                    W[p + n] = W[p + n][self._qk_perm.to(W[p + n].device)]
            blk = _lib.VitBlock(*[g(p + n).data_ptr() for n in (
                "norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")])
            check(L.vlm_vit_set_block(h, i, C.byref(blk)), "vit_set_block")
        gl = _lib.VitGlobals(wp.data_ptr(), *[g("merger." + n).data_ptr() for n in (
            "ln_q.weight", "ln_q.bias", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias")])
        check(L.vlm_vit_set_globals(h, C.byref(gl)), "vit_set_globals")

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().vlm_vit_destroy(self._handle)
        except Exception:
            pass

    # ------------------------------------------------------------------ host tables
    def rot_pos_emb(self, grid_thw) -> torch.Tensor:
        """freqs [N, head_dim/2] fp32 (reference vision.py:219-255)."""
        g = np.asarray(grid_thw)
        pos = rot_pos_ids(g, self.spatial_merge_size)
        dim = self.head_dim // 2
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
        seq = torch.arange(int(g[:, 1:].max()), dtype=torch.float32)
        full = torch.outer(seq, inv_freq)
        return full[torch.from_numpy(pos)].reshape(pos.shape[0], -1)

    def _tables(self, grid_thw):
        key = tuple(map(tuple, np.asarray(grid_thw).tolist()))
        hit = self._tab_cache.get(key)
        if hit is None:
            freqs = self.rot_pos_emb(grid_thw)
            lens = []
            for t, h, w in key:
                lens += [h * w] * t
            cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            nqb = int(sum((l + 127) // 128 for l in lens))
            cs = _lib.h2d(torch.stack([torch.cos(freqs), torch.sin(freqs)]).contiguous(), self.device)   # ONE table
            hit = (cs[0], cs[1], _lib.h2d(cu, self.device), len(lens), nqb, int(len(set(lens)) == 1))
            if len(self._tab_cache) > 64:
                self._tab_cache.clear()
            self._tab_cache[key] = hit
        return hit

    # ------------------------------------------------------------------ forward
    def __call__(self, hidden_states: torch.Tensor, grid_thw, output_hidden_states: Optional[bool] = None):
        """hidden_states: pixel_values [N, C*T*ph*pw] (f32 or bf16, rows as the processor emits them:
        columns ordered (C, T, ph, pw)).  -> [N / merge^2, hidden_size] bf16."""
        from ... import ops

        c = self.config
        if not hidden_states.is_cuda:
            hidden_states = _lib.h2d(hidden_states, self.device)
        N = hidden_states.shape[0]
        # PatchEmbed (vision.py:93-101): astype(weight dtype) + zero pad of K (the channels-last move is folded
        # into the weight, see load_weights)
        if hidden_states.dtype != torch.float32:
            hidden_states = hidden_states.to(torch.float32)
        x = ops.cast_pad(hidden_states.contiguous(), self.patch_k)
        cos, sin, cu, nseg, nqb, uniform = self._tables(grid_thw)
        E, dev = c.embed_dim, self.device
        mm = c.spatial_merge_size ** 2
        bf = torch.bfloat16
        ws = {n: torch.empty(N, d, dtype=bf, device=dev) for n, d in
              (("x", E), ("xn", E), ("qkv", 3 * E), ("attn", E), ("mlp", self.mlp_hidden))}
        mrg = torch.empty(N // mm, E * mm, dtype=bf, device=dev)
        out = torch.empty(N // mm, c.hidden_size, dtype=bf, device=dev)
        a = _lib.VitArgs(x.data_ptr(), N, cos.data_ptr(), sin.data_ptr(), cu.data_ptr(), nseg, nqb, ws["x"].data_ptr(),
                         ws["xn"].data_ptr(), ws["qkv"].data_ptr(), ws["attn"].data_ptr(), ws["mlp"].data_ptr(),
                         mrg.data_ptr(), out.data_ptr(), uniform)
        check(_lib.lib().vlm_vit_forward(self._handle, C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "vit_forward")
        return out

    # ------------------------------------------------------------------ checkpoint key/layout fixups
    def sanitize(self, weights):
        """reference vision.py:292-310."""
        out = {}
        for k, v in weights.items():
            if "position_ids" in k:
                continue
            if "patch_embed.proj.weight" in k:
                out[k] = v if check_array_shape(v) else v.permute(0, 2, 3, 4, 1)
            else:
                out[k] = v
        return out
