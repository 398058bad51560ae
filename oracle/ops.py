"""Oracle primitives (TEST INFRASTRUCTURE - see oracle/__init__.py).

torch-CPU restatement of the MLX ops the reference hot path calls.  Every
function cites the reference call site (paths relative to
/root/reference/mlx_vlm/).

Accumulation / rounding policy ("typed graph" emulation).  The reference runs
every tensor in the checkpoint dtype T (bf16 for the benchmark) and MLX ops
round their result to T.  We emulate that on CPU by computing each op in fp32
from T inputs and rounding ONCE at the op boundary, i.e. at the points where
the reference materialises a T tensor:
  * nn.Linear / addmm ......... fp32 accumulate, +bias in fp32, one rounding
  * nn.LayerNorm (mx.fast) .... fp32 stats and affine, one rounding
  * nn.RMSNorm (mx.fast) ...... fp32 stats; y = T(w * T(x * rsqrt(ms+eps)))
                                (normalised value is cast to T before the
                                weight multiply, as HF Qwen2RMSNorm does)
  * GELU / SiLU / mul / add ... fp32 math per elementary op, rounded per op; python scalars are converted to T
                                before the op (MLX weak typing: 1.702 * x multiplies by T(1.702))
  * mx.fast.sdpa .............. fp32 scores+softmax, P*V in fp32, one rounding
  * M-RoPE fused path ......... fp32 angle/cos/sin/rotation, one rounding
                                (rope_utils.py:589-603,640-643)
  * logits .................... T;  logprobs = T(logits - T(logsumexp)) (ar.py:368)
With T = float32 every rounding is the identity and the functions are the
plain fp32 math, which is what is checked against HuggingFace.
The rounding points are pinned by executing the reference's own files over
oracle/mlx_shim (tests/test_oracle_ref_golden.py: bit-exact in bf16); what stays
unpinned is the accumulation order inside MLX's kernels (oracle/__init__.py), so
the HIP kernels are compared with the tolerances stated in tests/.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch

F32 = torch.float32


# --------------------------------------------------------------------------
# dense ops
# --------------------------------------------------------------------------
def _result_type(*tensors):
    """MLX type promotion of array operands (bf16 with f32 -> f32).  Same-dtype graphs are unaffected; it matters for
    models that let float32 pixel values into a bf16 network without a cast (llava_bunny)."""
    t = tensors[0].dtype
    for x in tensors[1:]:
        t = torch.promote_types(t, x.dtype)
    return t


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None):
    """nn.Linear: y = x @ W.T + b (models/qwen2_vl/vision.py:129-130,168-169;
    language.py:52-55; mlp.py:9-11).  fp32 accumulate, one rounding.  A quantized weight (oracle/quant.py::QW - what
    load_model's nn.quantize swaps in for a 4-bit checkpoint, utils.py:918-967) goes through its quantized matmul."""
    if hasattr(w, "wq"):
        return w.linear(x, b)
    y = x.to(F32) @ w.to(F32).T
    if b is not None:
        y = y + b.to(F32)
    return y.to(_result_type(x, w))


def layer_norm(x, w, b, eps: float = 1e-6):
    """nn.LayerNorm(eps=1e-6) (vision.py:109,180-181) -> mx.fast.layer_norm."""
    xf = x.to(F32)
    mean = xf.mean(-1, keepdim=True)
    var = ((xf - mean) ** 2).mean(-1, keepdim=True)
    y = (xf - mean) * torch.rsqrt(var + eps)
    y = y * w.to(F32) + b.to(F32)
    return y.to(x.dtype)


def rms_norm(x, w, eps: float = 1e-6):
    """nn.RMSNorm (language.py:130-133,168) -> mx.fast.rms_norm."""
    xf = x.to(F32)
    inv = torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps)
    xn = (xf * inv).to(x.dtype)
    return (w.to(F32) * xn.to(F32)).to(x.dtype)


def add(a, b):
    """residual add in T (vision.py:188-193; language.py:151-153)."""
    return (a.to(F32) + b.to(F32)).to(_result_type(a, b))


def _c(v: float, T):
    """a python scalar as MLX sees it next to an array of dtype T: converted to T first (weak typing)."""
    return torch.tensor(v, dtype=T).to(F32)


def gelu_fast(x):
    """nn.GELU(approx="fast") = x * sigmoid(1.702 * x) (vision.py:167; mlx.nn.gelu_fast_approx).
    Typed graph (pinned by running the reference over oracle/mlx_shim): T(1.702) * x -> T, sigmoid -> T, x * s -> T."""
    T = x.dtype
    xf = x.to(F32)
    t1 = (_c(1.702, T) * xf).to(T)
    sg = torch.sigmoid(t1.to(F32)).to(T)
    return (xf * sg.to(F32)).to(T)


def gelu_erf(x):
    """nn.GELU() = x * (1 + erf(x / sqrt(2))) / 2 (vision.py:112; mlx.nn.gelu), every elementary op rounded to T."""
    T = x.dtype
    xf = x.to(F32)
    t1 = (xf / _c(math.sqrt(2.0), T)).to(T)
    e = torch.erf(t1.to(F32)).to(T)
    o = (1.0 + e.to(F32)).to(T)
    m = (xf * o.to(F32)).to(T)
    return (m.to(F32) / 2.0).to(T)


def swiglu(gate, up):
    """swiglu = silu(gate) * x (activations.py:7-9); silu(x) = x * sigmoid(x).
    Typed graph: sigmoid -> T, x*sig -> T, *up -> T."""
    T = gate.dtype
    gf = gate.to(F32)
    sig = torch.sigmoid(gf).to(T)
    silu = (gf * sig.to(F32)).to(T)
    return (silu.to(F32) * up.to(F32)).to(T)


def sdpa(q, k, v, scale: float, causal: bool = False, q_offset: int = 0,
         key_mask: Optional[torch.Tensor] = None):
    """mx.fast.scaled_dot_product_attention (vision.py:154; base.py:366-373).

    q [B,Hq,Lq,D], k/v [B,Hkv,Lk,D] (GQA: Hq % Hkv == 0, query head h uses kv
    head h // (Hq//Hkv)).  `causal`: key j visible to query i iff
    j <= i + q_offset, q_offset = Lk - Lq (cache.py:24-42 create_causal_mask).
    `key_mask` [B,Lk] bool: False = padded key (cache.py:1071-1074).
    fp32 scores and softmax, fp32 P*V, one rounding to q.dtype."""
    B, Hq, Lq, D = q.shape
    Hkv = k.shape[1]
    rep = Hq // Hkv
    qf = q.to(F32)
    kf = k.to(F32).repeat_interleave(rep, dim=1)
    vf = v.to(F32).repeat_interleave(rep, dim=1)
    s = (qf @ kf.transpose(-1, -2)) * scale
    Lk = k.shape[2]
    if causal:
        i = torch.arange(Lq)[:, None] + q_offset
        j = torch.arange(Lk)[None, :]
        s = s.masked_fill(~(j <= i), float("-inf"))
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ vf).to(_result_type(q, k, v))


# --------------------------------------------------------------------------
# rotary embeddings
# --------------------------------------------------------------------------
def vision_inv_freq(dim: int, theta: float = 10000.0):
    """VisionRotaryEmbedding (vision.py:53-65); dim = head_dim // 2."""
    return 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=F32) / dim))


def vision_rot_pos_ids(grid_thw: np.ndarray, merge: int = 2) -> np.ndarray:
    """VisionModel.rot_pos_emb integer part (vision.py:219-247): (h, w) index of
    every patch in merge-window order.  -> int64 [N, 2]."""
    out = []
    for t, h, w in np.asarray(grid_thw).tolist():
        hpos = np.repeat(np.arange(h)[:, None], w, axis=1)
        wpos = np.repeat(np.arange(w)[None, :], h, axis=0)

        def mw(a):
            a = a.reshape(h // merge, merge, w // merge, merge)
            return a.transpose(0, 2, 1, 3).reshape(-1)

        hw = np.stack([mw(hpos), mw(wpos)], axis=-1)
        out.append(np.tile(hw, (t, 1)))
    return np.concatenate(out, axis=0).astype(np.int64)


def vision_rotary_freqs(grid_thw: np.ndarray, head_dim: int, merge: int = 2):
    """rot_pos_emb (vision.py:219-255): freqs[N, head_dim//2] fp32 =
    concat(freq(h_pos), freq(w_pos))."""
    pos = vision_rot_pos_ids(grid_thw, merge)
    inv = vision_inv_freq(head_dim // 2)
    max_grid = int(np.asarray(grid_thw)[:, 1:].max())
    seq = torch.arange(max_grid, dtype=F32)
    full = torch.outer(seq, inv)  # [max_grid, head_dim//4]
    f = full[torch.from_numpy(pos)]  # [N, 2, head_dim//4]
    return f.reshape(pos.shape[0], -1)


def apply_rotary_pos_emb_vision(x, freqs):
    """apply_rotary_pos_emb_vision (vision.py:35-50).  x [N, H, D] (T),
    freqs [N, D/2] fp32.  cos/sin tiled x2 along D; rotate_half; fp32 math."""
    cos = torch.cos(freqs)[:, None, :].repeat(1, 1, 2)
    sin = torch.sin(freqs)[:, None, :].repeat(1, 1, 2)
    xf = x.to(F32)
    d = x.shape[-1] // 2
    rot = torch.cat([-xf[..., d:], xf[..., :d]], dim=-1)
    return (xf * cos + rot * sin).to(x.dtype)


def mrope_inv_freq(dim: int, base: float):
    """compute_inv_freq (rope_utils.py:1042-1043)."""
    return 1.0 / (base ** (torch.arange(0, dim, 2).to(F32) / dim))


def chunked_position_selector(mrope_section: Sequence[int], freq_dim: int):
    """_chunked_position_selector (rope_utils.py:519-526): frequency index f ->
    position axis (0=t, 1=h, 2=w)."""
    sel = [0] * freq_dim
    off = mrope_section[0]
    for dim, length in enumerate(mrope_section[1:], start=1):
        for idx in range(off, min(off + length, freq_dim)):
            sel[idx] = dim
        off += length
    return torch.tensor(sel, dtype=torch.int64)


def mrope_apply(x, position_ids, inv_freq, selector, mode: str = "fused"):
    """M-RoPE, half-split pairing, style "chunked".

    x [B, H, L, D]; position_ids [3, B, L] or [B, L] (ints).
    mode "fused": the Metal kernel's numerics (rope_utils.py:567-651):
        angle = float(pos[sel[f]]) * inv_freq[f]; c, s in fp32;
        out[d] = T(x_d c - x_p s), out[d+D/2] = T(x_p c + x_d s).
    mode "fallback": the non-Metal path (rope_utils.py:1235-1241,1486-1504 with
        compute_dtype=None): cos/sin rounded to T first, rotation in T-typed
        ops (mul -> T, mul -> T, add -> T)."""
    B, H, L, D = x.shape
    half = D // 2
    pos = position_ids
    if pos.dim() == 2:
        ang = pos.to(F32)[..., None] * inv_freq  # [B, L, half]
    else:
        p = pos[selector]  # [half, B, L]
        ang = p.permute(1, 2, 0).to(F32) * inv_freq
    c = torch.cos(ang)[:, None]  # [B,1,L,half]
    s = torch.sin(ang)[:, None]
    if mode == "fused":
        xf = x.to(F32)
        xd, xp = xf[..., :half], xf[..., half:]
        return torch.cat([xd * c - xp * s, xp * c + xd * s], dim=-1).to(x.dtype)
    T = x.dtype
    c = torch.cat([c, c], -1).to(T)
    s = torch.cat([s, s], -1).to(T)
    rot = torch.cat([-x[..., half:], x[..., :half]], dim=-1)
    a = (x.to(F32) * c.to(F32)).to(T)
    b = (rot.to(F32) * s.to(F32)).to(T)
    return (a.to(F32) + b.to(F32)).to(T)


# --------------------------------------------------------------------------
# KV cache (models/cache.py:337-393 KVCache)
# --------------------------------------------------------------------------
class KVCache:
    """Contiguous [B, Hkv, S, D] K and V, 256-step growth (cache.py:338-367)."""

    step = 256

    def __init__(self):
        self.keys = None
        self.values = None
        self.offset = 0

    def update_and_fetch(self, keys, values):
        prev = self.offset
        if self.keys is None or (prev + keys.shape[2]) > self.keys.shape[2]:
            B, H, _, D = keys.shape
            n_steps = (self.step + keys.shape[2] - 1) // self.step
            new_k = torch.zeros(B, H, n_steps * self.step, D, dtype=keys.dtype)
            new_v = torch.zeros(B, H, n_steps * self.step, values.shape[3], dtype=values.dtype)
            if self.keys is not None:
                if prev % self.step != 0:
                    self.keys = self.keys[..., :prev, :]
                    self.values = self.values[..., :prev, :]
                self.keys = torch.cat([self.keys, new_k], dim=2)
                self.values = torch.cat([self.values, new_v], dim=2)
            else:
                self.keys, self.values = new_k, new_v
        self.offset += keys.shape[2]
        self.keys[..., prev:self.offset, :] = keys
        self.values[..., prev:self.offset, :] = values
        return self.keys[..., :self.offset, :], self.values[..., :self.offset, :]

    @property
    def state(self):
        return self.keys[..., :self.offset, :], self.values[..., :self.offset, :]

    def trim(self, n):
        n = min(self.offset, n)
        self.offset -= n
        return n


class RotatingKVCache:
    """RotatingKVCache (cache.py:442-625) as make_prompt_cache builds it for `max_kv_size` (cache.py:65-68: keep = 4): the
    first `keep` tokens stay for ever; a multi-token update (the prompt) is kept WHOLE (_update_concat on an empty cache; a later
    one would first put the buffer in temporal order and trim it to max_size - 1 + S); a one-token update (_update_in_place)
    grows the buffer up to max_size, cuts a longer buffer down to keep + the most recent max_size - keep, and once the write
    index reaches max_size wraps it to `keep` - the new token overwrites the oldest entry that is not a sink.
    `_idx` (the write index) is what the language models read as the cache offset (qwen2_vl/language.py:430: `c0._idx if
    hasattr(c0, "_idx")`): after the first wrap the rope position of a decoded token is its RING index, not its count."""

    def __init__(self, max_size: int, keep: int = 0):
        self.keep, self.max_size = keep, max_size
        self.keys = self.values = None
        self.offset = 0
        self._idx = 0

    def _temporal(self, v):
        if self._idx == v.shape[2]:
            return v
        if self._idx < self.offset:
            return torch.cat([v[..., :self.keep, :], v[..., self._idx:, :], v[..., self.keep:self._idx, :]], dim=2)
        return v[..., :self._idx, :]

    def _cut(self, n, v, append=None):
        parts = [v[..., :self.keep, :], v[..., n + self.keep:, :]] if n > 0 else [v]
        if append is not None:
            parts.append(append)
        return torch.cat(parts, dim=2)

    def update_and_fetch(self, keys, values):
        S = keys.shape[2]
        if S != 1:                                                      # _update_concat (cache.py:486-505)
            if self.keys is None:
                self.keys, self.values = keys, values
            else:
                self.keys, self.values = self._temporal(self.keys), self._temporal(self.values)
                self._idx = self.keys.shape[2]
                n = self._idx - self.max_size + 1
                self.keys, self.values = self._cut(n, self.keys, keys), self._cut(n, self.values, values)
            self.offset += S
            self._idx = self.keys.shape[2]
            return self.keys, self.values
        prev = self.offset                                              # _update_in_place (cache.py:507-547)
        if self.keys is None or (prev >= self.keys.shape[2] and self.keys.shape[2] < self.max_size):
            B, H, _, D = keys.shape
            grow = min(256, self.max_size - prev)
            zk = torch.zeros(B, H, grow, D, dtype=keys.dtype)
            zv = torch.zeros(B, H, grow, values.shape[3], dtype=values.dtype)
            self.keys = zk if self.keys is None else torch.cat([self.keys, zk], dim=2)
            self.values = zv if self.values is None else torch.cat([self.values, zv], dim=2)
            self._idx = prev
        n = self.keys.shape[2] - self.max_size
        if n > 0:
            self.keys, self.values = self._cut(n, self.keys), self._cut(n, self.values)
            self._idx = self.max_size
        if self._idx == self.max_size:
            self._idx = self.keep
        self.keys[..., self._idx:self._idx + 1, :] = keys
        self.values[..., self._idx:self._idx + 1, :] = values
        self.offset += 1
        self._idx += 1
        if self.offset < self.max_size:
            return self.keys[..., :self.offset, :], self.values[..., :self.offset, :]
        return self.keys, self.values

    def size(self):
        return min(self.offset, self.max_size)

    def is_trimmable(self):
        return self.offset < self.max_size

    def trim(self, n):
        n = min(self.offset, n)
        self.offset -= n
        self._idx -= n
        return n


# --------------------------------------------------------------------------
# sampling (generate/ar.py:368; sample_utils.py)
# --------------------------------------------------------------------------
def logprobs_from_logits(logits):
    """logprobs = logits - logsumexp(logits) in the logits dtype (ar.py:368)."""
    T = logits.dtype
    lse = torch.logsumexp(logits.to(F32), dim=-1, keepdim=True).to(T)
    return (logits.to(F32) - lse.to(F32)).to(T)


def argmax_first(x):
    """mx.argmax (sample_utils.py:63-64): lowest index among ties."""
    xf = x.to(F32)
    m = xf.max(dim=-1, keepdim=True).values
    idx = torch.arange(x.shape[-1]).expand_as(xf)
    big = torch.full_like(idx, x.shape[-1])
    return torch.where(xf == m, idx, big).min(dim=-1).values


def apply_top_k(logprobs, top_k: int):
    """_apply_top_k (sample_utils.py:169-175): keep the k largest, rest -inf.
    Ties at the k-th value: argpartition keeps an arbitrary subset; the oracle
    keeps the lowest indices (stable descending sort) and says so."""
    xf = logprobs.to(F32)
    order = torch.sort(-xf, dim=-1, stable=True).indices
    out = xf.clone()
    out.scatter_(-1, order[..., top_k:], float("-inf"))
    return out.to(logprobs.dtype)


def apply_top_p(logprobs, top_p: float):
    """apply_top_p (sample_utils.py:289-318): ascending sort, cumulative probs,
    keep tokens whose cumulative prob (inclusive) > 1 - top_p.  Typed graph (matters for bf16 log-probs, which is what
    generate_step passes: tests/golden/samplers_ref.npz): exp -> T, cumsum -> T (fp32 accumulate, every prefix rounded), and the
    python scalar 1 - top_p is converted to T before the comparison."""
    T = logprobs.dtype
    xf = logprobs.to(F32)
    probs = torch.exp(xf).to(T).to(F32)
    order = torch.sort(xf, dim=-1, stable=True).indices
    sp = torch.gather(probs, -1, order)
    cum = torch.cumsum(sp, dim=-1).to(T).to(F32)
    inv = torch.empty_like(order)
    inv.scatter_(-1, order, torch.arange(order.shape[-1]).expand_as(order))
    cum = torch.gather(cum, -1, inv)
    return torch.where(cum > _c(1 - top_p, T), xf, torch.full_like(xf, float("-inf"))).to(T)


def apply_min_p(logprobs, min_p: float, min_tokens_to_keep: int = 1):
    """_apply_min_p (sample_utils.py:266-286).  min_tokens_to_keep > 1: the k largest are never removed
    (mx.argpartition(kth=-k)[-k:]; ties at the k-th value: the shim's stable ascending sort keeps the HIGHEST indices)."""
    T = logprobs.dtype
    xf = logprobs.to(F32)
    top = xf.max(dim=-1, keepdim=True).values
    # typed graph: `top_logprobs + math.log(min_p)` is an op of the logprobs dtype - the python scalar is converted to T first
    # and the sum is rounded to T (found with bf16 inputs in tests/golden/samplers_ref.npz; fp32 inputs cannot tell)
    remove = xf < (top + _c(math.log(min_p), T)).to(T).to(F32)
    if min_tokens_to_keep > 1:
        keep = torch.sort(xf, dim=-1, stable=True).indices[..., -min_tokens_to_keep:]
        remove.scatter_(-1, keep, False)
    return torch.where(remove, torch.full_like(xf, float("-inf")), xf).to(logprobs.dtype)


def apply_top_n_sigma(logits, n_sigma: float):
    """_top_n_sigma (sample_utils.py:181-212): keep x >= max - n_sigma * std, statistics of the float32 copy (ddof 0)."""
    f = logits.to(F32)
    top = f.max(dim=-1, keepdim=True).values
    std = f.var(dim=-1, correction=0, keepdim=True).sqrt()
    thr = top - n_sigma * std
    return torch.where(f < thr, torch.full_like(f, float("-inf")), f).to(logits.dtype)


def apply_p_less(logits, temp: float):
    """apply_p_less (sample_utils.py:215-236): keep tokens whose probability under softmax(logits / temp) is at least the
    collision probability sum p^2.  Typed graph: logits * T(1 / temp) -> T, softmax -> T, p * p -> T, sum -> T."""
    T = logits.dtype
    s = (logits.to(F32) * _c(1.0 / temp, T)).to(T)
    probs = torch.softmax(s.to(F32), dim=-1).to(T)
    sq = (probs.to(F32) * probs.to(F32)).to(T)
    thr = sq.to(F32).sum(dim=-1, keepdim=True).to(T)
    lf = logits.to(F32)
    return torch.where(probs.to(F32) < thr.to(F32), torch.full_like(lf, float("-inf")), lf).to(T)


def apply_typical_p(logprobs, typical_p: float):
    """_typical_p (sample_utils.py:321-345): tokens in ascending order of |-logp - entropy| (stable), kept while the
    cumulative probability BEFORE them is below typical_p.  Typed graph: every elementary op rounds to T."""
    T = logprobs.dtype
    lf = logprobs.to(F32)
    p = torch.exp(lf).to(T)
    pl = (p.to(F32) * lf).to(T)
    ent = (-(pl.to(F32).sum(dim=-1, keepdim=True).to(T)).to(F32)).to(T)
    shifted = ((-lf) - ent.to(F32)).to(T).to(F32).abs()
    order = torch.sort(shifted, dim=-1, stable=True).indices
    sp = torch.gather(p.to(F32), -1, order)
    cum = torch.cumsum(sp, dim=-1).to(T)
    inv = torch.empty_like(order)
    inv.scatter_(-1, order, torch.arange(order.shape[-1]).expand_as(order))
    cum = torch.gather(cum, -1, inv)
    before = (cum.to(F32) - p.to(F32)).to(T)
    return torch.where(before.to(F32) < _c(typical_p, T), lf, torch.full_like(lf, float("-inf"))).to(T)


def apply_xtc(logits, apply: bool, xtc_threshold: float, special_tokens=()):
    """apply_xtc (sample_utils.py:348-376) for ONE row: when the draw says so (`apply`), every token whose probability is
    strictly above the SMALLEST probability that exceeds the threshold is removed - i.e. of the tokens above the threshold
    only the least likely survives - except the special tokens."""
    if not apply:
        return logits
    T = logits.dtype
    lf = logits.to(F32)
    probs = torch.softmax(lf, dim=-1).to(T).to(F32)
    cand = torch.where(probs > _c(xtc_threshold, T), probs, torch.full_like(probs, float("inf"))).min()
    mask = probs > cand
    if len(special_tokens):
        mask[..., list(special_tokens)] = False
    return torch.where(mask, torch.full_like(lf, float("-inf")), lf).to(T)


def xtc_draw(seed: int, step: int, row: int = 0) -> float:
    """The uniform draw of OUR xtc (the reference's is mx.random.uniform, sample_utils.py:374 - MLX's stream is not
    reproducible outside MLX): the counter hash at an index no vocabulary entry has (csrc/sample.hip)."""
    return float(hash_uniform(seed, step, row, np.array([0xFFFFFFFF], dtype=np.uint64))[0])


def sampler_filters(logprobs, temp: float, top_p: float = 0.0, min_p: float = 0.0, min_tokens_to_keep: int = 1, top_k: int = 0,
                    top_n_sigma: float = 0.0, p_less: bool = False, typical_p: float = 1.0, xtc_probability: float = 0.0,
                    xtc_threshold: float = 0.0, xtc_special_tokens=(), seed: int = 0, step: int = 0):
    """The filter chain of make_sampler's closure (sample_utils.py:66-89) in its order, on [B, V] log-probs -> the filtered
    log-probs that categorical_sampling receives.  xtc: one row (its minimum runs over the whole array)."""
    x = logprobs
    if top_n_sigma > 0.0:
        x = apply_top_n_sigma(x, top_n_sigma)
    if p_less:
        x = apply_p_less(x, temp)
    if 0.0 < typical_p < 1.0:
        x = apply_typical_p(x, typical_p)
    if 0.0 < top_p < 1.0:
        x = apply_top_p(x, top_p)
    if min_p != 0.0:
        x = apply_min_p(x, min_p, min_tokens_to_keep)
    if xtc_probability > 0.0:
        assert x.shape[0] == 1
        # np.float32 comparison: the kernel compares the fp32 draw with float32(xtc_probability)
        x = apply_xtc(x, not (np.float32(xtc_draw(seed, step, 0)) > np.float32(xtc_probability)), xtc_threshold, xtc_special_tokens)
    if top_k > 0:
        x = apply_top_k(x, top_k)
    return x


def apply_logits_processors(logits, tokens, logit_bias=None, repetition_penalty=None, repetition_context_size=20,
                            presence_penalty=None, presence_context_size=20, frequency_penalty=None,
                            frequency_context_size=20):
    """make_logits_processors applied in its order (sample_utils.py:92-146): logit_bias (129-134), repetition penalty
    (390-422: sign-aware, once per distinct token of the last `context` fed tokens), presence penalty (425-450: minus p
    once per distinct token), frequency penalty (453-475: minus p per OCCURRENCE, `.at[].subtract`).  generate_step
    feeds `tokens` = prompt + every token fed back so far (ar.py:360-364).  Typed graph: every step rounds to the logits
    dtype; a python-float penalty is weak-typed (rounded to that dtype first); duplicate updates of `.at[]` apply one
    after the other.  logits [B, V] -> new tensor."""
    x = logits.clone()
    T = x.dtype
    toks = [int(t) for t in np.asarray(tokens).reshape(-1)]

    def c(v):
        return torch.tensor(v, dtype=T)

    if logit_bias:
        for k, v in logit_bias.items():
            x[:, int(k)] = (x[:, int(k)] + c(float(v))).to(T)
    if not toks:
        return x
    if repetition_penalty is not None and repetition_penalty != 0:
        for t in dict.fromkeys(toks[-repetition_context_size:]):
            col = x[:, t]
            x[:, t] = torch.where(col < 0, (col * c(repetition_penalty)).to(T), (col / c(repetition_penalty)).to(T))
    if presence_penalty is not None and presence_penalty != 0:
        for t in dict.fromkeys(toks[-presence_context_size:]):
            x[:, t] = (x[:, t] - c(presence_penalty)).to(T)
    if frequency_penalty is not None and frequency_penalty != 0:
        for t in toks[-frequency_context_size:]:
            x[:, t] = (x[:, t] - c(frequency_penalty)).to(T)
    return x


def hash_uniform(seed: int, step: int, row: int, idx: np.ndarray) -> np.ndarray:
    """Counter-based uniform in (0,1) used by OUR categorical sampler (the MLX
    RNG stream, sample_utils.py:385-387 mx.random.categorical, is not
    reproducible outside MLX - SURVEY.md §7.6).  32-bit mix of
    (seed, step, row, idx); restated bit-for-bit by the HIP kernel."""
    M = np.uint64(0xFFFFFFFF)
    x = (np.uint64(seed) ^ np.uint64(0x9E3779B9)) & M
    x = (x + (np.uint64(step) + np.uint64(1)) * np.uint64(0x85EBCA6B)) & M
    x = (x ^ ((np.uint64(row) + np.uint64(1)) * np.uint64(0xC2B2AE35))) & M
    x = (x + idx.astype(np.uint64) * np.uint64(0x27D4EB2F)) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    return ((x >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def categorical_gumbel(logprobs, temp: float, seed: int, step: int, row: int = 0):
    """categorical_sampling(logits, temp) = categorical(logits / temp)
    (sample_utils.py:385-387) by the Gumbel-max trick (which is what
    mx.random.categorical does) with OUR hash RNG.  logprobs [V]."""
    x = logprobs.to(F32).numpy() * np.float32(1.0 / temp)
    u = hash_uniform(seed, step, row, np.arange(x.shape[-1]))
    g = -np.log(-np.log(u.astype(np.float32))).astype(np.float32)
    z = (x + g).astype(np.float32)
    return int(np.argmax(z))


# ---------------------------------------------------------------------------------------------- synthetic weights
_NORMAL_POOL = []
_SCALED_POOLS = {}


def _normal_pool() -> torch.Tensor:
    """2^26 N(0, 1) floats, drawn once per process from numpy SFC64 streams (one per 4 Mi block, thread pool)."""
    if _NORMAL_POOL:
        return _NORMAL_POOL[0]
    import os
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    n, BLK = 1 << 26, 1 << 22
    buf = np.empty(n, dtype=np.float32)

    def fill(i):                                    # pure numpy: the GIL is released while drawing
        rng = np.random.Generator(np.random.SFC64(np.random.SeedSequence([0xC0FFEE, i])))
        rng.standard_normal(BLK, dtype=np.float32, out=buf[i * BLK:(i + 1) * BLK])

    try:
        workers = len(os.sched_getaffinity(0))
    except AttributeError:
        workers = os.cpu_count() or 1
    with ThreadPoolExecutor(max_workers=max(1, min(16, workers))) as ex:
        list(ex.map(fill, range(n // BLK)))
    _NORMAL_POOL.append(torch.from_numpy(buf))
    return _NORMAL_POOL[0]


def fast_normal(shape, seed, std: float, dtype=torch.bfloat16) -> torch.Tensor:
    """N(0, std^2) values for the BIG tensors of a full-size synthetic checkpoint (test infrastructure: 7-8 B parameters
    through torch.randn's single-threaded generator take two minutes per model on the GPU box's clock).  Every tensor is
    a run of windows of ONE 2^26-element normal pool, starting at an offset hashed from `seed` (an int or a tuple of
    ints) and jumping by a prime at every wrap, scaled and rounded per tensor - deterministic, seconds per model.
    Tensors therefore share values at different alignments; rows, layers and matrices still all differ, which is what
    a parity test needs (the oracle and the engine read the same checkpoint)."""
    pool = _normal_pool()
    P = pool.numel()
    n = 1
    for d in shape:
        n *= int(d)
    key = list(seed) if isinstance(seed, (tuple, list)) else [int(seed)]
    h = 0x9E3779B97F4A7C15
    for k in key:
        h = ((h ^ (int(k) & 0xFFFFFFFFFFFFFFFF)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        h ^= h >> 31
    off = h % P
    sk = (float(std), dtype)
    if sk not in _SCALED_POOLS:                    # the pool at this scale and dtype, once: tensors are then plain copies
        _SCALED_POOLS[sk] = (pool * float(std)).to(dtype)
    sp = _SCALED_POOLS[sk]
    out = torch.empty(n, dtype=dtype)
    pos = 0
    while pos < n:
        take = min(n - pos, P - off)
        out[pos:pos + take].copy_(sp[off:off + take])
        pos += take
        off = (off + take + 1000003) % P
    return out.reshape(*shape)
