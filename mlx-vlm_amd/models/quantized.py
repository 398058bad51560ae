"""MLX affine-quantized checkpoints (4 bits, group 64) - the loader half of the reference's `nn.quantize(...)` block in
`load_model` (mlx_vlm/utils.py:918-967): a Linear / Embedding is quantized iff the checkpoint holds `<path>.scales`; its
`weight` is then uint32 [out, in / 8] (8 nibbles per word, little end first), `scales` / `biases` [out, in / 64] in the
model dtype, and w = scales * q + biases.

Here the packed tensors are kept as they are and only re-laid-out for the HIP kernels (csrc/gemv_w4.hip): the q words
unchanged, scale and bias of a group fused into one uint32 (scale bf16 | bias bf16 << 16).  Row operations the engine
needs (q/k/v concatenation, gate/up interleave, embedding lookup) act on whole rows, so they commute with the packing."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

GROUP, BITS = 64, 4


@dataclass
class QuantW:
    wq: torch.Tensor      # int32 [N, K / 8]   (uint32 bit patterns)
    sb: torch.Tensor      # int32 [N, K / 64]  scale bf16 | bias bf16 << 16

    @property
    def shape(self):
        return (self.wq.shape[0], self.wq.shape[1] * 8)

    def to(self, device):
        return QuantW(self.wq.to(device).contiguous(), self.sb.to(device).contiguous())

    def rows(self, idx):
        return QuantW(self.wq[idx].contiguous(), self.sb[idx].contiguous())


def check_quantization(q: Optional[dict]):
    """config["quantization"] (utils.py:916): only MLX's affine 4-bit / group-64 mode is built"""
    if not q:
        return
    if int(q.get("bits", 4)) != BITS or int(q.get("group_size", 64)) != GROUP or q.get("mode", "affine") != "affine":
        raise NotImplementedError(f"quantization {q}: only affine, bits=4, group_size=64 is built (SURVEY section 8f.2)")


def has_scales(W: Dict[str, torch.Tensor], path: str) -> bool:
    """the reference's class predicate (utils.py:961): `f"{p}.scales" in weights`"""
    return f"{path}.scales" in W


def _bf16_bits(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).contiguous().view(torch.int16).to(torch.int32) & 0xFFFF


def take(W: Dict[str, torch.Tensor], path: str) -> QuantW:
    """`path`.weight / .scales / .biases -> QuantW (host tensors)"""
    wq, sc, bi = W[path + ".weight"], W[path + ".scales"], W[path + ".biases"]
    if wq.dtype not in (torch.int32, torch.uint32):
        raise ValueError(f"{path}.weight: expected uint32 words, got {wq.dtype}")
    if wq.dtype == torch.uint32:
        wq = wq.view(torch.int32)
    N, K = wq.shape[0], wq.shape[1] * 8
    if sc.shape != (N, K // GROUP) or bi.shape != sc.shape:
        raise ValueError(f"{path}: scales / biases {tuple(sc.shape)} do not match 4-bit words {tuple(wq.shape)} at group {GROUP}")
    sb = _bf16_bits(sc).to(torch.int64) | (_bf16_bits(bi).to(torch.int64) << 16)
    sb = torch.where(sb >= 2 ** 31, sb - 2 ** 32, sb)       # the uint32 bit pattern held in an int32
    return QuantW(wq.contiguous(), sb.to(torch.int32).contiguous())


def cat_rows(parts) -> QuantW:
    return QuantW(torch.cat([p.wq for p in parts], 0).contiguous(), torch.cat([p.sb for p in parts], 0).contiguous())


def interleave_rows(a: QuantW, b: QuantW) -> QuantW:
    """rows a0, b0, a1, b1, ... (gate / up -> SwiGLU epilogue layout)"""
    N = a.wq.shape[0]
    return QuantW(torch.stack([a.wq, b.wq], 1).reshape(2 * N, -1).contiguous(),
                  torch.stack([a.sb, b.sb], 1).reshape(2 * N, -1).contiguous())


def dequantize_bf16(q: QuantW, device) -> torch.Tensor:
    """the whole matrix as bf16 on the device (vlm_dequant_w4) - towers that run on the bf16 GEMMs"""
    from .. import ops

    q = q.to(device)
    return ops.dequant_w4(q.wq, q.sb)
