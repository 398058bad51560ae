"""MLX affine weight quantization restated on the CPU (test infrastructure - see oracle/__init__.py).

The reference quantizes / loads 4-bit checkpoints through the un-vendored `mlx` runtime: `nn.quantize(model, group_size,
bits, class_predicate)` in `load_model` (mlx_vlm/utils.py:918-967) swaps every Linear whose `<path>.scales` exists in the
checkpoint for `nn.QuantizedLinear`, whose forward is `mx.quantized_matmul(x, weight, scales, biases, transpose=True,
group_size, bits)` (+ bias), and `nn.Embedding` for `nn.QuantizedEmbedding` (`mx.dequantize` of the gathered rows; tied
lm_head = `as_linear` = the same quantized matmul).  None of that arithmetic is in /root/reference; it is restated here
from MLX 0.32's published semantics - "parity unpinned" applies (no reference golden can be produced without mlx):

  layout    weight uint32 [out, in * bits / 32]: element k of a row sits in word k // (32 / bits), bit field
            (k % (32 / bits)) * bits - little end first; scales / biases [out, in / group_size] in the model dtype
  dequant   w[o, k] = scales[o, k // gs] * q[o, k] + biases[o, k // gs]
  quantize  per group: w_max, w_min; scale = max((w_max - w_min) / (2^bits - 1), 1e-7), signed so that the edge of larger
            magnitude lands exactly on an integer (edge = |w_min| > |w_max| ? w_min : w_max; q0 = round(edge / scale);
            scale = edge / q0 if q0 != 0; bias = edge, or 0 when q0 == 0); q = clip(round((w - bias) / scale), 0, 2^bits-1)
  matmul    y = x . dequant(w)^T accumulated in fp32, rounded once to x's dtype (policy of oracle/ops.py::linear)
"""
from __future__ import annotations

import numpy as np
import torch

F32 = torch.float32


def quantize_affine(w: torch.Tensor, group_size: int = 64, bits: int = 4):
    """mx.quantize(w, group_size, bits) -> (wq uint32 [N, K * bits / 32] as int64-safe torch.int32 bit pattern, scales,
    biases) with scales / biases in w's dtype."""
    assert w.dim() == 2 and w.shape[1] % group_size == 0 and 32 % bits == 0
    N, K = w.shape
    n_bins = (1 << bits) - 1
    g = w.to(F32).reshape(N, K // group_size, group_size)
    w_max, w_min = g.max(-1).values, g.min(-1).values
    mask = w_min.abs() > w_max.abs()
    scales = torch.clamp((w_max - w_min) / n_bins, min=1e-7)
    scales = torch.where(mask, scales, -scales)
    edge = torch.where(mask, w_min, w_max)
    q0 = torch.round(edge / scales)
    scales = torch.where(q0 != 0, edge / q0, scales)
    biases = torch.where(q0 == 0, torch.zeros_like(edge), edge)
    # scales / biases are materialised in the weight dtype; the integers are computed from the fp32 values
    q = torch.clamp(torch.round((g - biases[..., None]) / scales[..., None]), 0, n_bins).to(torch.int64).reshape(N, K)
    per = 32 // bits
    q = q.reshape(N, K // per, per)
    shifts = (torch.arange(per, dtype=torch.int64) * bits)
    words = (q << shifts).sum(-1)                                   # < 2^32
    wq = words.to(torch.int64).numpy().astype(np.uint32)
    return torch.from_numpy(wq.view(np.int32)), scales.to(w.dtype), biases.to(w.dtype)


def unpack(wq: torch.Tensor, bits: int = 4) -> torch.Tensor:
    """uint32 words (int32 bit pattern) [N, W] -> integers [N, W * 32 / bits] (int64)"""
    per = 32 // bits
    u = torch.from_numpy(wq.numpy().view(np.uint32).astype(np.int64))
    shifts = (torch.arange(per, dtype=torch.int64) * bits)
    return ((u[..., None] >> shifts) & ((1 << bits) - 1)).reshape(wq.shape[0], -1)


def dequantize(wq, scales, biases, group_size: int = 64, bits: int = 4, dtype=None) -> torch.Tensor:
    """mx.dequantize: scales * q + biases per group, fp32 arithmetic, one rounding to the scales' dtype"""
    q = unpack(wq, bits).to(F32)
    N, K = q.shape
    w = q.reshape(N, K // group_size, group_size) * scales.to(F32)[..., None] + biases.to(F32)[..., None]
    return w.reshape(N, K).to(dtype or scales.dtype)


def quantized_linear(x, wq, scales, biases, bias=None, group_size: int = 64, bits: int = 4, w_f32=None):
    """nn.QuantizedLinear: mx.quantized_matmul(x, w, scales, biases, transpose=True) (+ bias).  fp32 accumulation over
    x (its dtype) times the fp32 dequantized weight, one rounding to x's dtype; the bias add is a second typed op.
    w_f32: the fp32 affine weights if the caller keeps them (QW.keep_f32 - same values, computed once)."""
    w = w_f32 if w_f32 is not None else dequantize(wq, scales, biases, group_size, bits, dtype=F32)
    y = (x.to(F32) @ w.T).to(x.dtype)
    if bias is not None:
        y = (y.to(F32) + bias.to(F32)).to(x.dtype)
    return y


class QW:
    """An MLX-quantized weight as the oracle's weight dicts carry it (what nn.QuantizedLinear / nn.QuantizedEmbedding hold)."""

    def __init__(self, wq, scales, biases, group_size: int = 64, bits: int = 4):
        self.wq, self.scales, self.biases, self.group_size, self.bits = wq, scales, biases, group_size, bits

    @property
    def dtype(self):
        return self.scales.dtype

    @property
    def shape(self):
        return (self.wq.shape[0], self.wq.shape[1] * 32 // self.bits)

    keep_f32 = False        # class-wide switch for long full-size runs: keep each weight's fp32 affine values after the first
                            # use instead of unpacking 4-bit words on every call (same values; 4 bytes per weight of host memory)

    def linear(self, x, bias=None):
        w = None
        if QW.keep_f32:
            w = self.__dict__.get("_f32")
            if w is None:
                w = self._f32 = dequantize(self.wq, self.scales, self.biases, self.group_size, self.bits, dtype=F32)
        return quantized_linear(x, self.wq, self.scales, self.biases, bias, self.group_size, self.bits, w_f32=w)

    def rows(self, idx):
        """nn.QuantizedEmbedding.__call__: mx.dequantize of the gathered rows"""
        idx = torch.as_tensor(idx, dtype=torch.long)
        flat = idx.reshape(-1)
        w = dequantize(self.wq[flat], self.scales[flat], self.biases[flat], self.group_size, self.bits)
        return w.reshape(*idx.shape, -1)


def quantize_checkpoint(W, predicate=None, group_size: int = 64, bits: int = 4):
    """What `mlx_vlm.convert` (nn.quantize on the model, then save) leaves in a 4-bit checkpoint: every 2-D `<path>.weight`
    accepted by `predicate(path, tensor)` becomes `<path>.weight` (uint32 words) + `<path>.scales` + `<path>.biases`.
    -> (checkpoint-style dict of tensors, oracle-style dict with QW objects under `<path>.weight`)"""
    ck, ow = {}, {}
    for k, v in W.items():
        path = k[: -len(".weight")] if k.endswith(".weight") else None
        if path is not None and v.dim() == 2 and v.shape[1] % group_size == 0 and (predicate is None or predicate(path, v)):
            wq, s, b = quantize_affine(v, group_size, bits)
            ck[path + ".weight"], ck[path + ".scales"], ck[path + ".biases"] = wq, s, b
            ow[k] = QW(wq, s, b, group_size, bits)
        else:
            ck[k] = v
            ow[k] = v
    return ck, ow


# ---------------------------------------------------------------------------------------------- quantized KV cache
# Reference: mlx_vlm/models/cache.py:233-334 (QuantizedKVCache), :415-423 (KVCache.to_quantized), models/base.py:260-302
# (quantized_scaled_dot_product_attention), generate/common.py:77-181 (maybe_quantize_kv_cache: the uniform path at the
# end - after EVERY forward of generate_step, ar.py:362, each layer's KVCache whose offset >= quantized_kv_start becomes
# a QuantizedKVCache).  mx.quantize / mx.quantized_matmul themselves are restated above ("parity unpinned").
def quantize_nd(x: torch.Tensor, group_size: int = 64, bits: int = 8):
    """mx.quantize over the LAST axis of [..., D] -> (wq int32 words [..., D * bits / 32], scales [..., D / gs], biases)"""
    lead, D = x.shape[:-1], x.shape[-1]
    wq, s, b = quantize_affine(x.reshape(-1, D), group_size, bits)
    return wq.reshape(*lead, -1), s.reshape(*lead, -1), b.reshape(*lead, -1)


def dequantize_nd(wq, scales, biases, group_size: int = 64, bits: int = 8, dtype=F32) -> torch.Tensor:
    lead = wq.shape[:-1]
    w = dequantize(wq.reshape(-1, wq.shape[-1]), scales.reshape(-1, scales.shape[-1]), biases.reshape(-1, biases.shape[-1]),
                   group_size, bits, dtype=dtype)
    return w.reshape(*lead, -1)


class QuantizedKVCache:
    """cache.py:233-334: keys / values held as (words, scales, biases); update_and_fetch quantizes the new rows with
    mx.quantize(group_size, bits) and returns everything up to the offset, still quantized.  (The 256-step growth of the
    backing arrays has no effect on values and is not restated.)"""

    def __init__(self, group_size: int = 64, bits: int = 8):
        self.keys = None
        self.values = None
        self.offset = 0
        self.group_size, self.bits = group_size, bits

    def update_and_fetch(self, keys, values):
        nk, nv = quantize_nd(keys, self.group_size, self.bits), quantize_nd(values, self.group_size, self.bits)
        if self.keys is None:
            self.keys, self.values = nk, nv
        else:
            self.keys = tuple(torch.cat([a[..., : self.offset, :], b], dim=-2) for a, b in zip(self.keys, nk))
            self.values = tuple(torch.cat([a[..., : self.offset, :], b], dim=-2) for a, b in zip(self.values, nv))
        self.offset += keys.shape[-2]
        return self.keys, self.values

    def trim(self, n):
        n = min(self.offset, n)
        self.offset -= n
        return n


def to_quantized(cache, group_size: int = 64, bits: int = 8) -> QuantizedKVCache:
    """KVCache.to_quantized (cache.py:415-423): the rows up to the offset are quantized as they are"""
    q = QuantizedKVCache(group_size, bits)
    q.offset = cache.offset
    if cache.keys is not None:
        k, v = cache.keys[..., : cache.offset, :], cache.values[..., : cache.offset, :]
        q.keys, q.values = quantize_nd(k, group_size, bits), quantize_nd(v, group_size, bits)
    return q


def should_quantize_kv_layer(layer_idx: int, num_layers: int) -> bool:
    """models/cache.py:8-21 - the BATCH policy (generate/ar.py:842-858 `_make_cache`): with kv_bits set every layer's batch
    cache is quantised from the start, except the last layer of a stack deeper than 2 (kept in the model dtype)."""
    return True if num_layers <= 2 else layer_idx < num_layers - 1


def maybe_quantize_kv_cache(prompt_cache: list, quantized_kv_start: int, kv_group_size: int, kv_bits, batch_policy: bool = False) -> None:
    """generate/common.py:170-181 (uniform scheme): in place, every layer whose plain cache has reached the start offset.
    batch_policy: only the layers should_quantize_kv_layer names (the batched path's rule, generate/ar.py:842-858) - used
    by the tests of BatchGenerator(kv_bits=8), whose rows are quantised at their join (quantized_kv_start = 0)."""
    if kv_bits is None:
        return
    n = len(prompt_cache)
    for i, c in enumerate(prompt_cache):
        if batch_policy and not should_quantize_kv_layer(i, n):
            continue
        if not isinstance(c, QuantizedKVCache) and c.offset >= quantized_kv_start:
            prompt_cache[i] = to_quantized(c, kv_group_size, int(kv_bits))


def quantized_sdpa(q: torch.Tensor, q_keys, q_values, scale: float, causal: bool = False, group_size: int = 64, bits: int = 8):
    """quantized_scaled_dot_product_attention (base.py:260-302) as its typed graph: queries *= scale (a typed multiply:
    rounded to the query dtype), scores = quantized_matmul(q, K^T) (fp32 accumulation over the fp32-dequantized keys, ONE
    rounding to the query dtype), causal mask as where(mask, scores, finfo.min), softmax(precise=True) (fp32 inside, result
    in the scores' dtype), out = quantized_matmul(P, V) (fp32 accumulation, one rounding).  q [B, Hq, L, D]; q_keys /
    q_values tuples over [B, Hkv, S, ...]."""
    B, Hq, L, D = q.shape
    Hkv = q_keys[0].shape[1]
    rep = Hq // Hkv
    dt = q.dtype
    # `queries *= scale` with a python float: MLX's weak scalar takes the ARRAY's dtype first (bf16(128 ** -0.5) =
    # 0.08837890625, not 0.0883883...), then one typed multiply - pinned by tests/golden/kvquant_ref.npz
    qs = (q.to(F32) * torch.tensor(scale, dtype=dt).to(F32)).to(dt)
    kf = dequantize_nd(*q_keys, group_size, bits).repeat_interleave(rep, dim=1)          # [B, Hq, S, D] fp32
    vf = dequantize_nd(*q_values, group_size, bits).repeat_interleave(rep, dim=1)
    scores = (qs.to(F32) @ kf.transpose(-1, -2)).to(dt)
    if causal:
        S = kf.shape[2]
        i = torch.arange(S - L, S)[:, None]
        j = torch.arange(S)[None, :]
        scores = torch.where(i >= j, scores, torch.full_like(scores, torch.finfo(dt).min))
    p = torch.softmax(scores.to(F32), dim=-1).to(dt)
    return (p.to(F32) @ vf).to(dt)
