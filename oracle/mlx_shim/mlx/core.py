"""mlx.core stand-in (TEST INFRASTRUCTURE - see oracle/mlx_shim/README.md).

A torch-CPU implementation of the subset of Apple MLX's `mlx.core` API that the Qwen2-VL hot path of the
reference calls, so that the reference's OWN Python files can be imported and executed in a container where
`mlx` (pinned by the reference: mlx==0.32.0, uv.lock:1094-1095) cannot be installed.  It restates MLX's
published op semantics:

  * arrays are typed; every op returns its result rounded to the result dtype (torch computes bf16
    element-wise ops in fp32 and rounds once - the same "typed graph" contract);
  * python scalars do not promote an array's dtype - they are first converted to it (1.702 * x with x bf16
    multiplies by bf16(1.702)); int defaults to int32, float to float32;
  * matmul / addmm accumulate in fp32 and round once;
  * mx.fast.{rms_norm, layer_norm, scaled_dot_product_attention} compute in fp32 internally
    (rms_norm: y = w * T(x * rsqrt(mean(x^2) + eps)); layer_norm: one rounding; sdpa: fp32 scores, softmax and
    P.V, one rounding; "causal" is lower-right aligned);
  * mx.metal.is_available() is False, so the reference takes its pure-MLX M-RoPE path
    (models/rope_utils.py:654-689), not the fused Metal kernel.

Never imported by the product or by the GPU tests; only tests/golden/make_golden_ref.py uses it.
"""
from __future__ import annotations

import builtins
import contextlib
import math
from typing import Any, Optional, Sequence

import numpy as np
import torch

# ---------------------------------------------------------------------------------------------- dtypes
Dtype = torch.dtype
float32, float16, bfloat16, float64 = torch.float32, torch.float16, torch.bfloat16, torch.float64
int8, int16, int32, int64 = torch.int8, torch.int16, torch.int32, torch.int64
uint8, uint16, uint64 = torch.uint8, torch.uint16, torch.uint64


class _SizedDtype:
    """mx.uint32 with MLX's `Dtype.size` (the reference computes `8 * mx.uint32.size // bits`, models/cache.py:249);
    torch.dtype objects cannot carry extra attributes.  Equal to, hashes like and unwraps to torch.uint32."""

    def __init__(self, t, size):
        self._torch, self.size = t, size

    def __eq__(self, o):
        return o is self or o == self._torch

    def __hash__(self):
        return hash(self._torch)

    def __repr__(self):
        return "mlx.core.uint32"


uint32 = _SizedDtype(torch.uint32, 4)


def _dt(d):
    return d._torch if isinstance(d, _SizedDtype) else d


bool_ = torch.bool
complex64 = torch.complex64
inf, nan, pi, e, newaxis = float("inf"), float("nan"), math.pi, math.e, None
floating = "floating"
inexact = "inexact"
integer = "integer"


def issubdtype(dt, kind):
    if kind in (floating, inexact):
        return dt in (float32, float16, bfloat16, float64)
    if kind == integer:
        return dt in (int8, int16, int32, int64, uint8, uint16, uint32, uint64)
    return dt == kind


def _py(v):
    """nested python structure with arrays replaced by python values"""
    if isinstance(v, array):
        return v._t.tolist()
    if isinstance(v, (list, tuple)):
        return [_py(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def _to_tensor(val, dtype=None) -> torch.Tensor:
    dtype = _dt(dtype)
    if isinstance(val, array):
        t = val._t
    elif isinstance(val, torch.Tensor):
        t = val
    elif isinstance(val, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(val))
        if t.dtype == torch.float64 and dtype is None:
            t = t.to(torch.float32)
    else:
        v = _py(val)
        t = torch.tensor(v)
        if dtype is None:
            if t.dtype == torch.int64:
                t = t.to(torch.int32)
            elif t.dtype == torch.float64:
                t = t.to(torch.float32)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t


def _u(x):
    """unwrap for torch calls: arrays -> tensors, everything else unchanged"""
    return x._t if isinstance(x, array) else x


def _idx(i):
    if isinstance(i, tuple):
        return tuple(_idx(j) for j in i)
    if isinstance(i, array):
        return i._t.long() if i._t.dtype not in (torch.bool, torch.int64) else i._t
    if isinstance(i, list):
        return [_idx(j) for j in i]
    return i


def _axes(axis):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return tuple(int(a) for a in axis)
    return int(axis)


class _At:
    def __init__(self, arr):
        self._a, self._i = arr, None

    def __getitem__(self, i):
        self._i = i
        return self

    def _apply(self, v, fn):
        t = self._a._t.clone()
        i = self._i if isinstance(self._i, tuple) else (self._i,)
        lead, last = i[:-1], _u(i[-1])
        for x in lead:
            if not (isinstance(x, slice) and x == slice(None)):
                raise NotImplementedError("shim .at[]: only [..., indices] forms are restated")
        idx = torch.as_tensor(last).reshape(-1).tolist()
        vt = _u(v)
        vt = torch.as_tensor(vt).to(t.dtype).reshape(-1) if not isinstance(vt, (int, float)) else torch.tensor([vt], dtype=t.dtype)
        for k, j in enumerate(idx):
            t[..., j] = fn(t[..., j], vt[k % vt.numel()])
        return array(t)

    def add(self, v):
        return self._apply(v, torch.add)

    def subtract(self, v):
        return self._apply(v, torch.sub)


class array:
    __slots__ = ("_t",)

    def __init__(self, val=0, dtype=None):
        self._t = _to_tensor(val, dtype)

    # ---- introspection
    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def dtype(self):
        return self._t.dtype

    @property
    def ndim(self):
        return self._t.dim()

    @property
    def size(self):
        return self._t.numel()

    @property
    def itemsize(self):
        return self._t.element_size()

    @property
    def nbytes(self):
        return self._t.numel() * self._t.element_size()

    @property
    def T(self):
        return array(self._t.permute(*reversed(range(self._t.dim()))))

    def __len__(self):
        return self._t.shape[0]

    def __iter__(self):
        for i in range(self._t.shape[0]):
            yield array(self._t[i])

    def __repr__(self):
        return f"array({self._t.tolist()}, dtype={str(self._t.dtype).replace('torch.', '')})"

    def item(self):
        return self._t.item()

    def tolist(self):
        return self._t.tolist()

    def __bool__(self):
        return bool(self._t.item())

    def __int__(self):
        return int(self._t.item())

    def __float__(self):
        return float(self._t.item())

    def __index__(self):
        return int(self._t.item())

    def __array__(self, dtype=None, copy=None):
        t = self._t
        if t.dtype == torch.bfloat16:
            t = t.to(torch.float32)
        a = t.numpy()
        return a.astype(dtype) if dtype is not None else a

    __hash__ = None

    # ---- shape ops
    def astype(self, dtype, stream=None):
        return array(self._t.to(_dt(dtype)))

    def reshape(self, *shape, stream=None):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return array(self._t.reshape(tuple(int(_u(s)) if not isinstance(s, int) else s for s in shape)))

    def transpose(self, *axes, stream=None):
        if len(axes) == 1 and isinstance(axes[0], (tuple, list)):
            axes = tuple(axes[0])
        if not axes:
            return self.T
        return array(self._t.permute(*axes))

    def swapaxes(self, a, b):
        return array(self._t.transpose(a, b))

    def moveaxis(self, src, dst):
        return array(torch.movedim(self._t, src, dst))

    def flatten(self, start_axis=0, end_axis=-1):
        return array(torch.flatten(self._t, start_axis, end_axis))

    def squeeze(self, axis=None):
        return squeeze(self, axis)

    def split(self, indices_or_sections, axis=0):
        return split(self, indices_or_sections, axis)

    def view(self, dtype):
        return array(self._t.view(_dt(dtype)))

    # ---- reductions
    def sum(self, axis=None, keepdims=False):
        return sum(self, axis, keepdims)

    def mean(self, axis=None, keepdims=False):
        return mean(self, axis, keepdims)

    def max(self, axis=None, keepdims=False):
        return max(self, axis, keepdims)

    def min(self, axis=None, keepdims=False):
        return min(self, axis, keepdims)

    def prod(self, axis=None, keepdims=False):
        return prod(self, axis, keepdims)

    def all(self, axis=None, keepdims=False):
        return all(self, axis, keepdims)

    def any(self, axis=None, keepdims=False):
        return any(self, axis, keepdims)

    def argmax(self, axis=None, keepdims=False):
        return argmax(self, axis, keepdims)

    def argmin(self, axis=None, keepdims=False):
        return argmin(self, axis, keepdims)

    def cumsum(self, axis=None, reverse=False, inclusive=True):
        return cumsum(self, axis, reverse=reverse, inclusive=inclusive)

    def exp(self):
        return exp(self)

    def log(self):
        return log(self)

    def sqrt(self):
        return sqrt(self)

    def rsqrt(self):
        return rsqrt(self)

    def square(self):
        return square(self)

    def abs(self):
        return abs(self)

    def round(self, decimals=0):
        return array(torch.round(self._t, decimals=decimals))

    # ---- indexing
    def __getitem__(self, i):
        return array(self._t[_idx(i)])

    @property
    def at(self):
        """`a.at[idx].add(v)` / `.subtract(v)`: MLX's scatter-update forms.  Unlike `a[idx] += v`, EVERY update is applied,
        duplicates in `idx` included (mlx.core.array.at docs); the updates are applied one after the other in the
        array's dtype (scatter with a reduction = read-modify-write per update), i.e. k duplicates round k times."""
        return _At(self)

    def __setitem__(self, i, v):
        v = _u(v)
        if isinstance(v, torch.Tensor) and v.dtype != self._t.dtype:
            v = v.to(self._t.dtype)
        self._t[_idx(i)] = v

    # ---- arithmetic
    def _bin(self, other, fn, rev=False):
        o = _u(other)
        if isinstance(o, (list, tuple, np.ndarray)):
            o = _to_tensor(o)
        elif isinstance(o, (int, float)) and not isinstance(o, bool) and self._t.dtype in (bfloat16, float16):
            # MLX turns a python scalar into an array of the OTHER operand's dtype before the op (weak typing):
            # 1.702 * x with x bf16 multiplies by bf16(1.702) = 1.703125
            o = torch.tensor(o, dtype=self._t.dtype)
        a, b = (o, self._t) if rev else (self._t, o)
        return array(fn(a, b))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, torch.true_divide)
    def __rtruediv__(self, o): return self._bin(o, torch.true_divide, True)
    def __floordiv__(self, o): return self._bin(o, torch.floor_divide)
    def __rfloordiv__(self, o): return self._bin(o, torch.floor_divide, True)
    def __mod__(self, o): return self._bin(o, torch.remainder)
    def __pow__(self, o): return self._bin(o, torch.pow)
    def __rpow__(self, o): return self._bin(o, lambda a, b: torch.pow(torch.as_tensor(a, dtype=b.dtype if b.is_floating_point() else torch.float32), b), True)
    def __matmul__(self, o): return matmul(self, o)
    def __rmatmul__(self, o): return matmul(o, self)
    def __neg__(self): return array(-self._t)
    def __invert__(self): return array(~self._t)
    def __abs__(self): return array(torch.abs(self._t))
    def __eq__(self, o): return self._bin(o, torch.eq)
    def __ne__(self, o): return self._bin(o, torch.ne)
    def __lt__(self, o): return self._bin(o, torch.lt)
    def __le__(self, o): return self._bin(o, torch.le)
    def __gt__(self, o): return self._bin(o, torch.gt)
    def __ge__(self, o): return self._bin(o, torch.ge)
    def __and__(self, o): return self._bin(o, torch.bitwise_and)
    def __or__(self, o): return self._bin(o, torch.bitwise_or)
    def __xor__(self, o): return self._bin(o, torch.bitwise_xor)
    def __lshift__(self, o): return self._bin(o, torch.bitwise_left_shift)
    def __rshift__(self, o): return self._bin(o, torch.bitwise_right_shift)


def _a(x, dtype=None):
    return x if isinstance(x, array) and dtype is None else array(x, dtype)


# ---------------------------------------------------------------------------------------------- creation
def arange(start, stop=None, step=1, dtype=None, stream=None):
    start, stop, step = _py(start), _py(stop), _py(step)
    if stop is None:
        start, stop = 0, start
    if dtype is None:
        isf = builtins.any(isinstance(v, float) for v in (start, stop, step))
        dtype = float32 if isf else int32
    return array(torch.arange(start, stop, step, dtype=torch.float32 if dtype in (bfloat16, float16) else dtype).to(dtype))


def zeros(shape, dtype=float32, stream=None):
    return array(torch.zeros(tuple(_py(shape)) if not isinstance(shape, int) else (shape,), dtype=_dt(dtype)))


def ones(shape, dtype=float32, stream=None):
    return array(torch.ones(tuple(_py(shape)) if not isinstance(shape, int) else (shape,), dtype=_dt(dtype)))


def full(shape, vals, dtype=None, stream=None):
    v = _to_tensor(vals, dtype)
    return array(torch.broadcast_to(v, tuple(_py(shape)) if not isinstance(shape, int) else (shape,)).clone())


def zeros_like(a, stream=None):
    return array(torch.zeros_like(_a(a)._t))


def ones_like(a, stream=None):
    return array(torch.ones_like(_a(a)._t))


def linspace(start, stop, num=50, dtype=float32, stream=None):
    return array(torch.linspace(_py(start), _py(stop), num, dtype=torch.float32).to(dtype))


def eye(n, m=None, k=0, dtype=float32, stream=None):
    return array(torch.eye(n, m or n, dtype=dtype))


def tri(n, m=None, k=0, dtype=float32, stream=None):
    return array(torch.tril(torch.ones(n, m or n), diagonal=k).to(dtype))


def tril(a, k=0, stream=None):
    return array(torch.tril(_a(a)._t, diagonal=k))


def triu(a, k=0, stream=None):
    return array(torch.triu(_a(a)._t, diagonal=k))


# ---------------------------------------------------------------------------------------------- shape
def reshape(a, shape, stream=None):
    return _a(a).reshape(*shape)


def transpose(a, axes=None, stream=None):
    return _a(a).transpose(*(axes or ()))


def swapaxes(a, axis1, axis2, stream=None):
    return _a(a).swapaxes(axis1, axis2)


def moveaxis(a, source, destination, stream=None):
    return _a(a).moveaxis(source, destination)


def flatten(a, start_axis=0, end_axis=-1, stream=None):
    return _a(a).flatten(start_axis, end_axis)


def unflatten(a, axis, shape, stream=None):
    return array(torch.unflatten(_a(a)._t, axis, tuple(shape)))


def expand_dims(a, axis, stream=None):
    t = _a(a)._t
    if isinstance(axis, (list, tuple)):
        for ax in sorted(ax if ax >= 0 else ax + t.dim() + len(axis) for ax in axis):
            t = t.unsqueeze(ax)
        return array(t)
    return array(t.unsqueeze(axis))


def squeeze(a, axis=None, stream=None):
    t = _a(a)._t
    if axis is None:
        return array(t.squeeze())
    return array(t.squeeze(_axes(axis)))


def broadcast_to(a, shape, stream=None):
    return array(torch.broadcast_to(_a(a)._t, tuple(_py(shape))))


def broadcast_arrays(*arrs, stream=None):
    return [array(t) for t in torch.broadcast_tensors(*[_a(x)._t for x in arrs])]


def concatenate(arrays, axis=0, stream=None):
    ts = [_a(x)._t for x in arrays]
    dt = ts[0].dtype
    for t in ts[1:]:
        dt = torch.promote_types(dt, t.dtype)
    ts = [t.to(dt) for t in ts]
    if axis is None:
        return array(torch.cat([t.reshape(-1) for t in ts]))
    return array(torch.cat(ts, dim=axis))


def stack(arrays, axis=0, stream=None):
    ts = [_a(x)._t for x in arrays]
    dt = ts[0].dtype
    for t in ts[1:]:
        dt = torch.promote_types(dt, t.dtype)
    return array(torch.stack([t.to(dt) for t in ts], dim=axis))


def split(a, indices_or_sections, axis=0, stream=None):
    t = _a(a)._t
    ios = _py(indices_or_sections)
    if isinstance(ios, int):
        return [array(x) for x in torch.tensor_split(t, ios, dim=axis)]
    return [array(x) for x in torch.tensor_split(t, [int(i) for i in ios], dim=axis)]


def tile(a, reps, stream=None):
    t = _a(a)._t
    reps = (reps,) if isinstance(reps, int) else tuple(_py(reps))
    return array(torch.tile(t, reps))


def repeat(a, repeats, axis=None, stream=None):
    t = _a(a)._t
    r = int(_py(repeats))
    if axis is None:
        return array(t.reshape(-1).repeat_interleave(r))
    if t.dim() == 0:
        return array(t.reshape(1).repeat_interleave(r))
    return array(t.repeat_interleave(r, dim=axis))


def pad(a, pad_width, mode="constant", constant_values=0, stream=None):
    t = _a(a)._t
    pw = _py(pad_width)
    if isinstance(pw, int):
        pw = [(pw, pw)] * t.dim()
    elif isinstance(pw[0], int):
        pw = [tuple(pw)] if t.dim() == 1 or len(pw) == 2 and t.dim() == 1 else [tuple(pw)] * t.dim()
    flat = []
    for lo, hi in reversed(pw):
        flat += [lo, hi]
    return array(torch.nn.functional.pad(t, flat, mode="constant", value=_py(constant_values)))


class finfo:
    """mx.finfo(dtype): min / max / eps of a floating type"""

    def __init__(self, dtype):
        fi = torch.finfo(_dt(dtype))
        self.min, self.max, self.eps, self.dtype = fi.min, fi.max, fi.eps, dtype


def where(cond, x, y, stream=None):
    c = _a(cond)._t
    if c.dtype != torch.bool:
        c = c != 0
    xt, yt = _u(x), _u(y)
    if not isinstance(xt, torch.Tensor) and not isinstance(yt, torch.Tensor):
        xt = _to_tensor(xt)
    if not isinstance(xt, torch.Tensor):
        xt = torch.as_tensor(xt, dtype=yt.dtype)
    if not isinstance(yt, torch.Tensor):
        yt = torch.as_tensor(yt, dtype=xt.dtype)
    return array(torch.where(c, xt, yt))


def take(a, indices, axis=None, stream=None):
    t, i = _a(a)._t, _a(indices)._t.long()
    if axis is None:
        return array(t.reshape(-1)[i])
    axis %= t.dim()
    r = torch.index_select(t, axis, i.reshape(-1))
    return array(r.reshape(tuple(t.shape[:axis]) + tuple(i.shape) + tuple(t.shape[axis + 1:])))


def take_along_axis(a, indices, axis=None, stream=None):
    t, i = _a(a)._t, _a(indices)._t.long()
    if axis is None:
        return array(t.reshape(-1)[i.reshape(-1)])
    axis %= t.dim()
    shp = list(torch.broadcast_shapes(tuple(1 if d == axis else n for d, n in enumerate(t.shape)),
                                      tuple(1 if d == axis else n for d, n in enumerate(i.shape))))
    ts, is_ = list(shp), list(shp)
    ts[axis], is_[axis] = t.shape[axis], i.shape[axis]
    return array(torch.gather(t.expand(ts), axis, i.expand(is_)))


def put_along_axis(a, indices, values, axis=None, stream=None):
    t, i = _a(a)._t.clone(), _a(indices)._t.long()
    v = _a(values)._t.to(t.dtype)
    v = torch.broadcast_to(v, i.shape)
    return array(t.scatter(axis, i, v))


# ---------------------------------------------------------------------------------------------- math
def _unary(fn):
    def f(a, stream=None):
        return array(fn(_a(a)._t))
    return f


def _f(fn):
    """float op: ints are promoted to float32 like MLX"""
    def f(a, stream=None):
        t = _a(a)._t
        if not t.is_floating_point():
            t = t.to(torch.float32)
        return array(fn(t))
    return f


exp, expm1, log, log2, log10, log1p = _f(torch.exp), _f(torch.expm1), _f(torch.log), _f(torch.log2), _f(torch.log10), _f(torch.log1p)
sin, cos, tan, tanh, sinh, cosh = _f(torch.sin), _f(torch.cos), _f(torch.tan), _f(torch.tanh), _f(torch.sinh), _f(torch.cosh)
arctan, arcsin, arccos = _f(torch.atan), _f(torch.asin), _f(torch.acos)
sqrt, rsqrt, sigmoid, erf, erfinv = _f(torch.sqrt), _f(torch.rsqrt), _f(torch.sigmoid), _f(torch.erf), _f(torch.erfinv)
square, abs, negative, sign = _unary(torch.square), _unary(torch.abs), _unary(torch.neg), _unary(torch.sign)
floor, ceil = _unary(torch.floor), _unary(torch.ceil)
logical_not = _unary(torch.logical_not)
isnan, isinf, isfinite = _unary(torch.isnan), _unary(torch.isinf), _unary(torch.isfinite)
stop_gradient = _unary(lambda t: t)


def round(a, decimals=0, stream=None):
    return array(torch.round(_a(a)._t, decimals=decimals))


def _binary(name):
    def f(a, b, stream=None):
        a = _a(a) if not isinstance(b, array) or isinstance(a, array) else a
        if isinstance(a, array):
            return getattr(a, name)(b)
        return getattr(_a(b), name.replace("__", "__r", 1))(a)
    return f


add, subtract, multiply, divide = _binary("__add__"), _binary("__sub__"), _binary("__mul__"), _binary("__truediv__")
floor_divide, remainder, power = _binary("__floordiv__"), _binary("__mod__"), _binary("__pow__")
equal, not_equal, less, less_equal = _binary("__eq__"), _binary("__ne__"), _binary("__lt__"), _binary("__le__")
greater, greater_equal = _binary("__gt__"), _binary("__ge__")


def _tt(a, b):
    a, b = _u(a), _u(b)
    if not isinstance(a, torch.Tensor):
        a = torch.as_tensor(a, dtype=b.dtype if isinstance(b, torch.Tensor) else None)
    if not isinstance(b, torch.Tensor):
        b = torch.as_tensor(b, dtype=a.dtype)
    return a, b


def maximum(a, b, stream=None):
    return array(torch.maximum(*_tt(a, b)))


def minimum(a, b, stream=None):
    return array(torch.minimum(*_tt(a, b)))


def logical_and(a, b, stream=None):
    return array(torch.logical_and(*_tt(a, b)))


def logical_or(a, b, stream=None):
    return array(torch.logical_or(*_tt(a, b)))


def clip(a, a_min, a_max, stream=None):
    return array(torch.clamp(_a(a)._t, _py(a_min), _py(a_max)))


def matmul(a, b, stream=None):
    ta, tb = _a(a)._t, _a(b)._t
    dt = torch.promote_types(ta.dtype, tb.dtype)
    if dt in (bfloat16, float16):
        return array((ta.to(torch.float32) @ tb.to(torch.float32)).to(dt))     # fp32 accumulate, one rounding
    return array(ta.to(dt) @ tb.to(dt))


def addmm(c, a, b, alpha=1.0, beta=1.0, stream=None):
    tc, ta, tb = _a(c)._t, _a(a)._t, _a(b)._t
    dt = torch.promote_types(torch.promote_types(ta.dtype, tb.dtype), tc.dtype)
    y = alpha * (ta.to(torch.float32) @ tb.to(torch.float32)) + beta * tc.to(torch.float32)
    return array(y.to(dt))


def outer(a, b, stream=None):
    return array(torch.outer(_a(a)._t.reshape(-1), _a(b)._t.reshape(-1)))


def inner(a, b, stream=None):
    return array(torch.inner(_a(a)._t, _a(b)._t))


def einsum(subscripts, *operands, stream=None):
    return array(torch.einsum(subscripts, *[_a(o)._t for o in operands]))


def _red(fn, int_ok=True):
    def f(a, axis=None, keepdims=False, stream=None):
        t = _a(a)._t
        ax = _axes(axis)
        if ax is None:
            r = fn(t.reshape(-1), 0, False)
            return array(r.reshape((1,) * t.dim()) if keepdims else r)
        if isinstance(ax, tuple):
            r = t
            for d in sorted((d % t.dim() for d in ax), reverse=True):
                r = fn(r, d, keepdims)
            return array(r)
        return array(fn(t, ax, keepdims))
    return f


def _sum(t, d, k):
    if t.dtype == torch.bool:
        t = t.to(torch.int32)
    if t.dtype in (bfloat16, float16):
        return t.to(torch.float32).sum(d, keepdim=k).to(t.dtype)     # fp32 accumulate, one rounding
    return t.sum(d, keepdim=k).to(t.dtype)                           # ints keep their dtype (torch widens to int64)


sum = _red(_sum)
mean = _red(lambda t, d, k: t.to(torch.float32).mean(d, keepdim=k).to(t.dtype if t.is_floating_point() else torch.float32))
max = _red(lambda t, d, k: t.max(d, keepdim=k).values)
min = _red(lambda t, d, k: t.min(d, keepdim=k).values)
prod = _red(lambda t, d, k: t.prod(d, keepdim=k))
all = _red(lambda t, d, k: t.bool().all(d, keepdim=k))
any = _red(lambda t, d, k: t.bool().any(d, keepdim=k))


def var(a, axis=None, keepdims=False, ddof=0, stream=None):
    t = _a(a)._t
    return array(t.to(torch.float32).var(_axes(axis), correction=ddof, keepdim=keepdims).to(t.dtype))


def std(a, axis=None, keepdims=False, ddof=0, stream=None):
    t = _a(a)._t
    return array(t.to(torch.float32).var(_axes(axis), correction=ddof, keepdim=keepdims).sqrt().to(t.dtype))


def logsumexp(a, axis=None, keepdims=False, stream=None):
    t = _a(a)._t
    ax = _axes(axis)
    if ax is None:
        ax = tuple(range(t.dim()))
    return array(torch.logsumexp(t.to(torch.float32), ax, keepdim=keepdims).to(t.dtype))


def softmax(a, axis=-1, precise=False, stream=None):
    t = _a(a)._t
    return array(torch.softmax(t.to(torch.float32), _axes(axis)).to(t.dtype))


def argmax(a, axis=None, keepdims=False, stream=None):
    t = _a(a)._t
    if axis is None:
        return array(t.reshape(-1).argmax().to(torch.int32))
    # first occurrence on ties (MLX / numpy convention); torch.argmax does not guarantee it for all dtypes
    m = t.max(axis, keepdim=True).values
    idx = torch.arange(t.shape[axis]).reshape([-1 if d == axis % t.dim() else 1 for d in range(t.dim())])
    first = torch.where(t == m, idx, t.shape[axis]).min(axis, keepdim=keepdims).values
    return array(first.to(torch.int32))


def argmin(a, axis=None, keepdims=False, stream=None):
    return argmax(array(-_a(a)._t.to(torch.float32)), axis, keepdims)


def argsort(a, axis=-1, stream=None):
    return array(torch.argsort(_a(a)._t.to(torch.float32) if _a(a)._t.dtype == bfloat16 else _a(a)._t, dim=axis, stable=True).to(torch.int32))


def sort(a, axis=-1, stream=None):
    return array(torch.sort(_a(a)._t, dim=axis, stable=True).values)


def argpartition(a, kth, axis=-1, stream=None):
    return argsort(a, axis)       # a full sort is a valid partition


def partition(a, kth, axis=-1, stream=None):
    return sort(a, axis)


def topk(a, k, axis=-1, stream=None):
    return array(torch.topk(_a(a)._t, k, dim=axis).values.flip(axis))


def cumsum(a, axis=None, reverse=False, inclusive=True, stream=None):
    t = _a(a)._t
    if axis is None:
        t, axis = t.reshape(-1), 0
    src = t.to(torch.float32) if t.dtype in (bfloat16, float16) else (t.to(torch.int32) if t.dtype == torch.bool else t)
    if reverse:
        src = src.flip(axis)
    r = torch.cumsum(src, axis)
    if not inclusive:
        r = r - src
    if reverse:
        r = r.flip(axis)
    return array(r.to(t.dtype if t.dtype != torch.bool else torch.int32))


def cumprod(a, axis=None, stream=None):
    t = _a(a)._t
    return array(torch.cumprod(t, axis if axis is not None else 0))


def array_equal(a, b, equal_nan=False, stream=None):
    return array(torch.equal(_a(a)._t, _a(b)._t))


def allclose(a, b, rtol=1e-5, atol=1e-8, equal_nan=False, stream=None):
    return array(torch.allclose(_a(a)._t.to(torch.float32), _a(b)._t.to(torch.float32), rtol=rtol, atol=atol))


def meshgrid(*arrs, sparse=False, indexing="xy", stream=None):
    return [array(t) for t in torch.meshgrid(*[_a(x)._t for x in arrs], indexing=indexing)]


def contiguous(a, allow_col_major=False, stream=None):
    return array(_a(a)._t.contiguous())


def as_strided(a, shape=None, strides=None, offset=0, stream=None):
    return array(torch.as_strided(_a(a)._t, shape, strides, offset))


# ---- affine quantization (mx.quantize / mx.dequantize / mx.quantized_matmul, mode="affine")
# The arithmetic lives in the un-vendored mlx runtime; it is restated ONCE, in oracle/quant.py, from MLX's published
# algorithm (per group: signed scale, the edge of larger magnitude lands on an integer, 1e-7 floor, little-end packing;
# dequantize = scale * q + bias in fp32, one rounding; quantized_matmul = fp32 accumulation over the fp32 affine weights,
# one rounding to x's dtype) and stays "parity unpinned".  What these entry points make possible is running the
# REFERENCE'S OWN callers of them (models/cache.py QuantizedKVCache / to_quantized, models/base.py quantized SDPA,
# generate/common.py's switch-over, utils.py's nn.quantize load path) unmodified, which pins the graph around them.
def _words(t):
    return t.view(torch.int32) if t.dtype == torch.uint32 else t.to(torch.int32)


def quantize(w, group_size=64, bits=4, mode="affine", stream=None):
    if mode != "affine":
        raise NotImplementedError("mlx shim: only affine quantization is restated")
    from oracle import quant as _q

    wq, s, b = _q.quantize_nd(_a(w)._t, int(group_size), int(bits))
    return array(wq.contiguous().view(torch.uint32)), array(s), array(b)          # (array.dtype of the words == mx.uint32)


def dequantize(w, scales, biases=None, group_size=64, bits=4, mode="affine", dtype=None, stream=None):
    if mode != "affine" or biases is None:
        raise NotImplementedError("mlx shim: only affine quantization is restated")
    from oracle import quant as _q

    s = _a(scales)._t
    return array(_q.dequantize_nd(_words(_a(w)._t.contiguous()), s, _a(biases)._t, int(group_size), int(bits), dtype=dtype or s.dtype))


def quantized_matmul(x, w, scales, biases=None, transpose=True, group_size=64, bits=4, mode="affine", stream=None):
    if mode != "affine" or biases is None:
        raise NotImplementedError("mlx shim: only affine quantization is restated")
    from oracle import quant as _q

    xt = _a(x)._t
    wf = _q.dequantize_nd(_words(_a(w)._t.contiguous()), _a(scales)._t, _a(biases)._t, int(group_size), int(bits), dtype=torch.float32)
    y = xt.to(torch.float32) @ (wf.transpose(-1, -2) if transpose else wf)          # batch dims broadcast as mx.matmul's do
    return array(y.to(xt.dtype))


# ---------------------------------------------------------------------------------------------- runtime no-ops
def eval(*args, **kwargs):
    return None


async_eval = eval


def synchronize(stream=None):
    return None


def compile(fun=None, inputs=None, outputs=None, shapeless=False):
    if fun is None:
        return lambda f: f
    return fun


def checkpoint(fun):
    return fun


def disable_compile():
    return None


enable_compile = disable_compile


class Device:
    def __init__(self, type="cpu", index=0):
        self.type, self.index = type, index

    def __eq__(self, o):
        return isinstance(o, Device) and o.type == self.type

    def __repr__(self):
        return f"Device({self.type}, {self.index})"


cpu, gpu = Device("cpu"), Device("gpu")


class Stream:
    def __init__(self, device=cpu):
        self.device = device


def default_device():
    return cpu


def set_default_device(d):
    return None


def default_stream(device=None):
    return Stream()


def new_stream(device=None):
    return Stream()


def set_default_stream(s):
    return None


def new_thread_local_stream(device=None):
    return Stream()


def concat(arrays, axis=0, stream=None):
    return concatenate(arrays, axis)


@contextlib.contextmanager
def stream(s=None):
    yield


def clear_cache():
    return None


def get_peak_memory():
    return 0


def get_active_memory():
    return 0


def get_cache_memory():
    return 0


def reset_peak_memory():
    return None


def set_wired_limit(n):
    return 0


def set_cache_limit(n):
    return 0


def set_memory_limit(n, relaxed=True):
    return 0


def device_info():
    return {"max_recommended_working_set_size": 1 << 40, "memory_size": 1 << 40, "architecture": "shim"}


class _Metal:
    @staticmethod
    def is_available():
        return False

    @staticmethod
    def device_info():
        return device_info()

    get_peak_memory = staticmethod(get_peak_memory)
    get_active_memory = staticmethod(get_active_memory)
    get_cache_memory = staticmethod(get_cache_memory)
    clear_cache = staticmethod(clear_cache)
    reset_peak_memory = staticmethod(reset_peak_memory)
    set_wired_limit = staticmethod(set_wired_limit)
    set_cache_limit = staticmethod(set_cache_limit)
    set_memory_limit = staticmethod(set_memory_limit)


metal = _Metal()


class _Cuda:
    @staticmethod
    def is_available():
        return False


cuda = _Cuda()


# ---------------------------------------------------------------------------------------------- mx.fast
class _Fast:
    @staticmethod
    def rms_norm(x, weight, eps, stream=None):
        t = _a(x)._t
        xf = t.to(torch.float32)
        inv = torch.rsqrt((xf * xf).mean(-1, keepdim=True) + eps)
        xn = (xf * inv).to(t.dtype)
        if weight is None:
            return array(xn)
        return array((_a(weight)._t.to(torch.float32) * xn.to(torch.float32)).to(t.dtype))

    @staticmethod
    def layer_norm(x, weight, bias, eps, stream=None):
        t = _a(x)._t
        xf = t.to(torch.float32)
        mu = xf.mean(-1, keepdim=True)
        var_ = ((xf - mu) ** 2).mean(-1, keepdim=True)
        y = (xf - mu) * torch.rsqrt(var_ + eps)
        if weight is not None:
            y = y * _a(weight)._t.to(torch.float32)
        if bias is not None:
            y = y + _a(bias)._t.to(torch.float32)
        return array(y.to(t.dtype))

    @staticmethod
    def scaled_dot_product_attention(q, k, v, *, scale, mask=None, sinks=None, stream=None):
        if sinks is not None:
            raise NotImplementedError("mlx shim: attention sinks")
        tq, tk, tv = _a(q)._t, _a(k)._t, _a(v)._t
        B, Hq, Lq, D = tq.shape
        Hkv, Lk = tk.shape[1], tk.shape[2]
        rep = Hq // Hkv
        qf = tq.to(torch.float32)
        kf = tk.to(torch.float32).repeat_interleave(rep, dim=1)
        vf = tv.to(torch.float32).repeat_interleave(rep, dim=1)
        s = (qf @ kf.transpose(-1, -2)) * scale
        if isinstance(mask, str):
            if mask != "causal":
                raise ValueError(mask)
            i = torch.arange(Lq)[:, None] + (Lk - Lq)
            j = torch.arange(Lk)[None, :]
            s = s.masked_fill(~(j <= i), float("-inf"))
        elif mask is not None:
            m = _a(mask)._t
            if m.dtype == torch.bool:
                s = s.masked_fill(~m, float("-inf"))
            else:
                s = s + m.to(torch.float32)
        p = torch.softmax(s, dim=-1)
        # mlx/fast.cpp: q, k, v are cast to result_type(q, k, v), which is also the output type
        out_t = torch.promote_types(torch.promote_types(tq.dtype, tk.dtype), tv.dtype)
        return array((p @ vf).to(out_t))

    @staticmethod
    def rope(x, dims, *, traditional, base, scale, offset, freqs=None, stream=None):
        t = _a(x)._t
        L = t.shape[-2]
        off = _py(offset)
        if freqs is not None:
            inv = 1.0 / _a(freqs)._t.to(torch.float32)
        else:
            inv = 1.0 / (base ** (torch.arange(0, dims, 2, dtype=torch.float32) / dims))
        if isinstance(off, list) and len(off) > 1:
            # mx.fast.rope: "offset (int or array): ... a scalar or a vector of B offsets, one for each example in the batch":
            # row b of x [B, ..., L, D] rotates at positions offset[b] + 0 .. L-1
            if len(off) != t.shape[0]:
                raise ValueError(f"mlx shim: rope offset vector of {len(off)} for a batch of {t.shape[0]}")
            pos = (torch.arange(L, dtype=torch.float32)[None, :] + torch.tensor([float(o) for o in off])[:, None]) * scale
            ang = pos[:, :, None] * inv[None, None, :]                           # [B, L, dims/2]
            ang = ang.reshape((t.shape[0],) + (1,) * (t.dim() - 3) + (L, inv.numel()))
        else:
            pos = (torch.arange(L, dtype=torch.float32) + float(off if not isinstance(off, list) else off[0])) * scale
            ang = pos[:, None] * inv[None, :]
        c, s_ = torch.cos(ang), torch.sin(ang)
        xf = t.to(torch.float32)
        rot, rest = xf[..., :dims], xf[..., dims:]
        if traditional:
            a, b = rot[..., 0::2], rot[..., 1::2]
            o = torch.stack([a * c - b * s_, b * c + a * s_], dim=-1).flatten(-2)
        else:
            a, b = rot[..., : dims // 2], rot[..., dims // 2:]
            o = torch.cat([a * c - b * s_, b * c + a * s_], dim=-1)
        return array(torch.cat([o, rest], dim=-1).to(t.dtype))

    @staticmethod
    def metal_kernel(*a, **k):
        raise RuntimeError("mlx shim: no Metal (mx.metal.is_available() is False)")


fast = _Fast()


# ---------------------------------------------------------------------------------------------- mx.random
class _Random:
    def __init__(self):
        self._g = torch.Generator().manual_seed(0)

    def seed(self, s):
        self._g.manual_seed(int(s))

    def key(self, s):
        return array([0, int(s)], uint32)

    def split(self, key, num=2):
        return [key] * num

    def uniform(self, low=0.0, high=1.0, shape=(), dtype=float32, key=None, stream=None):
        return array((torch.rand(tuple(shape), generator=self._g) * (high - low) + low).to(dtype))

    def normal(self, shape=(), dtype=float32, loc=0.0, scale=1.0, key=None, stream=None):
        return array((torch.randn(tuple(shape), generator=self._g) * scale + loc).to(dtype))

    def randint(self, low, high, shape=(), dtype=int32, key=None, stream=None):
        return array(torch.randint(int(low), int(high), tuple(shape), generator=self._g).to(dtype))

    def gumbel(self, shape=(), dtype=float32, key=None, stream=None):
        u = torch.rand(tuple(shape), generator=self._g)
        return array((-torch.log(-torch.log(u))).to(dtype))

    def categorical(self, logits, axis=-1, shape=None, num_samples=None, key=None, stream=None):
        t = _a(logits)._t.to(torch.float32)
        g = -torch.log(-torch.log(torch.rand(t.shape, generator=self._g)))
        return array((t + g).argmax(axis).to(torch.int32))

    @property
    def state(self):
        return [array(0)]


random = _Random()


class _Linalg:
    @staticmethod
    def norm(a, ord=None, axis=None, keepdims=False, stream=None):
        t = _a(a)._t
        return array(torch.linalg.norm(t.to(torch.float32), ord=ord, dim=_axes(axis), keepdim=keepdims).to(t.dtype))


linalg = _Linalg()


class distributed:
    """only the names the reference's signatures mention at import time (utils.py:1124)"""

    class Group:
        pass

    @staticmethod
    def init(*a, **k):
        raise NotImplementedError("mlx shim: mx.distributed is outside the pinned path")


def save_safetensors(*a, **k):
    raise NotImplementedError("mlx shim")


load = save = savez = save_safetensors


def __getattr__(name):
    raise AttributeError(f"mlx shim (oracle/mlx_shim): mlx.core.{name} is outside the pinned Qwen2-VL path")
