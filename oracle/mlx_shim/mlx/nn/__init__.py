"""mlx.nn stand-in (TEST INFRASTRUCTURE, see oracle/mlx_shim/README.md): Module + the layers the Qwen2-VL path
of the reference instantiates, with MLX's documented semantics (nn.Linear = x @ W.T + b via addmm, one rounding;
nn.GELU() exact erf, approx="fast" -> x * sigmoid(1.702 x), approx="precise"/"tanh" -> tanh form;
nn.Embedding.as_linear; Conv3d channels-last, weight [O, kD, kH, kW, I])."""
from __future__ import annotations

import math
from functools import partial  # noqa: F401

import torch

from .. import core as mx
from ..utils import tree_flatten, tree_unflatten


class Module:
    def __init__(self):
        self._training = False

    # mlx.nn.Module is a dict: arrays / containers / sub-modules assigned as attributes are stored as ITEMS, bypassing any
    # class-level property of the same name, and `__getattr__` serves them when normal lookup fails.  The reference relies
    # on that (models/idefics2/language.py: a `layers` property whose getter raises sits next to `self.layers = [...]`).
    def __setattr__(self, key, val):
        desc = getattr(type(self), key, None)
        if isinstance(desc, property) and desc.fset is None:
            self.__dict__[key] = val
        else:
            object.__setattr__(self, key, val)

    def __getattr__(self, key):
        d = object.__getattribute__(self, "__dict__")
        if key in d:
            return d[key]
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {key!r}")

    # ---- parameter tree
    def _items(self):
        for k, v in self.__dict__.items():
            if k.startswith("_"):
                continue
            yield k, v

    def parameters(self):
        def rec(v):
            if isinstance(v, Module):
                return v.parameters()
            if isinstance(v, (list, tuple)):
                r = [rec(x) for x in v]
                return r if any(x is not None for x in r) else None
            if isinstance(v, dict):
                r = {k: rec(x) for k, x in v.items()}
                return r if any(x is not None for x in r.values()) else None
            if isinstance(v, mx.array):
                return v
            return None

        out = {}
        for k, v in self._items():
            r = rec(v)
            if isinstance(r, mx.array) or (r is not None and len(r) > 0):
                out[k] = r
        return out

    trainable_parameters = parameters

    def children(self):
        return {k: v for k, v in self._items() if isinstance(v, (Module, list, dict))}

    def named_modules(self):
        out = [("", self)]

        def rec(prefix, v):
            if isinstance(v, Module):
                out.append((prefix, v))
                for k, c in v._items():
                    rec(f"{prefix}.{k}" if prefix else k, c)
            elif isinstance(v, (list, tuple)):
                for i, c in enumerate(v):
                    rec(f"{prefix}.{i}", c)
            elif isinstance(v, dict):
                for k, c in v.items():
                    rec(f"{prefix}.{k}", c)

        for k, c in self._items():
            rec(k, c)
        return out

    def modules(self):
        return [m for _, m in self.named_modules()]

    def leaf_modules(self):
        return {}

    def _set(self, key, value):
        parts = key.split(".")
        obj = self
        for p in parts[:-1]:
            obj = obj[int(p)] if isinstance(obj, (list, tuple)) else (obj[p] if isinstance(obj, dict) else getattr(obj, p))
        last = parts[-1]
        if isinstance(obj, list):
            obj[int(last)] = value
        elif isinstance(obj, dict):
            obj[last] = value
        else:
            setattr(obj, last, value)

    def _get(self, key):
        obj = self
        for p in key.split("."):
            obj = obj[int(p)] if isinstance(obj, (list, tuple)) else (obj[p] if isinstance(obj, dict) else getattr(obj, p))
        return obj

    def load_weights(self, file_or_weights, strict=True):
        weights = list(file_or_weights.items()) if isinstance(file_or_weights, dict) else list(file_or_weights)
        have = dict(tree_flatten(self.parameters()))
        if strict:
            new = {k for k, _ in weights}
            missing, extra = set(have) - new, new - set(have)
            if missing or extra:
                raise ValueError(f"load_weights(strict): missing {sorted(missing)[:5]} extra {sorted(extra)[:5]}")
        for k, v in weights:
            if k in have:
                if tuple(have[k].shape) != tuple(v.shape):
                    raise ValueError(f"shape mismatch for {k}: {have[k].shape} vs {v.shape}")
                self._set(k, v if isinstance(v, mx.array) else mx.array(v))
        return self

    def update(self, parameters, strict=True):
        for k, v in tree_flatten(parameters):
            self._set(k, v)
        return self

    def apply(self, map_fn, filter_fn=None):
        for k, v in tree_flatten(self.parameters()):
            self._set(k, map_fn(v))
        return self

    def set_dtype(self, dtype, predicate=None):
        return self.apply(lambda a: a.astype(dtype) if a.dtype in (mx.float32, mx.float16, mx.bfloat16) else a)

    def eval(self):
        self._training = False
        return self

    def train(self, mode=True):
        self._training = mode
        return self

    def freeze(self, *a, **k):
        return self

    unfreeze = freeze

    @property
    def training(self):
        return self._training

    @property
    def state(self):
        return self.__dict__


def _init(shape, scale):
    g = torch.Generator().manual_seed(sum(shape) * 7919 + len(shape))
    return mx.array((torch.rand(shape, generator=g) * 2 - 1) * scale)


class Linear(Module):
    def __init__(self, input_dims, output_dims, bias=True):
        super().__init__()
        s = math.sqrt(1.0 / input_dims)
        self.weight = _init((output_dims, input_dims), s)
        if bias:
            self.bias = _init((output_dims,), s)

    def __call__(self, x):
        if "bias" in self.__dict__:
            return mx.addmm(self.bias, x, self.weight.T)
        return x @ self.weight.T


class Identity(Module):
    def __call__(self, x):
        return x


class Embedding(Module):
    def __init__(self, num_embeddings, dims):
        super().__init__()
        self.weight = _init((num_embeddings, dims), math.sqrt(1.0 / dims))

    def __call__(self, x):
        return self.weight[x]

    def as_linear(self, x):
        return x @ self.weight.T


class RMSNorm(Module):
    def __init__(self, dims, eps=1e-5):
        super().__init__()
        self.weight = mx.ones((dims,))
        self.eps = eps

    def __call__(self, x):
        return mx.fast.rms_norm(x, self.weight, self.eps)


class LayerNorm(Module):
    def __init__(self, dims, eps=1e-5, affine=True, bias=True):
        super().__init__()
        self.eps, self.dims = eps, dims
        if affine:
            self.weight = mx.ones((dims,))
            if bias:
                self.bias = mx.zeros((dims,))

    def __call__(self, x):
        return mx.fast.layer_norm(x, self.__dict__.get("weight"), self.__dict__.get("bias"), self.eps)


class _ConvNd(Module):
    nd = 2

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        nd = self.nd
        tup = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * nd
        self.kernel_size, self.stride, self.padding, self.dilation = tup(kernel_size), tup(stride), tup(padding), tup(dilation)
        self.groups = groups
        fan = in_channels * math.prod(self.kernel_size)
        self.weight = _init((out_channels, *self.kernel_size, in_channels // groups), math.sqrt(1.0 / fan))
        if bias:
            self.bias = mx.zeros((out_channels,))

    def __call__(self, x):
        nd = self.nd
        t = x._t
        perm_in = (0, nd + 1) + tuple(range(1, nd + 1))          # channels-last -> channels-first
        w = self.weight._t.permute((0, nd + 1) + tuple(range(1, nd + 1)))
        fn = {1: torch.nn.functional.conv1d, 2: torch.nn.functional.conv2d, 3: torch.nn.functional.conv3d}[nd]
        b = self.__dict__.get("bias")
        y = fn(t.permute(perm_in).to(torch.float32), w.to(torch.float32), None if b is None else b._t.to(torch.float32),
               stride=self.stride, padding=self.padding, dilation=self.dilation, groups=self.groups)
        y = y.permute((0,) + tuple(range(2, nd + 2)) + (1,))
        return mx.array(y.to(t.dtype))


class Conv1d(_ConvNd):
    nd = 1


class Conv2d(_ConvNd):
    nd = 2


class Conv3d(_ConvNd):
    nd = 3


# ---- activations (each elementary op rounds to the array dtype, as MLX's typed graph does)
def sigmoid(x):
    return mx.sigmoid(x)


def silu(x):
    return x * mx.sigmoid(x)


def relu(x):
    return mx.maximum(x, 0)


def softplus(x):
    return mx.log1p(mx.exp(x)) if hasattr(mx, "log1p") else mx.log(1 + mx.exp(x))


def gelu(x):
    return x * (1 + mx.erf(x / math.sqrt(2))) / 2


def gelu_approx(x):
    return 0.5 * x * (1 + mx.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x**3)))


def gelu_fast_approx(x):
    return x * mx.sigmoid(1.702 * x)


def tanh(x):
    return mx.tanh(x)


def softmax(x, axis=-1):
    return mx.softmax(x, axis=axis)


class GELU(Module):
    def __init__(self, approx="none"):
        super().__init__()
        self._act = {"none": gelu, "precise": gelu_approx, "tanh": gelu_approx, "fast": gelu_fast_approx}[approx]

    def __call__(self, x):
        return self._act(x)


class SiLU(Module):
    def __call__(self, x):
        return silu(x)


class ReLU(Module):
    def __call__(self, x):
        return relu(x)


class Tanh(Module):
    def __call__(self, x):
        return mx.tanh(x)


class Sigmoid(Module):
    def __call__(self, x):
        return mx.sigmoid(x)


class Dropout(Module):
    def __init__(self, p=0.5):
        super().__init__()

    def __call__(self, x):
        return x


class Sequential(Module):
    def __init__(self, *modules):
        super().__init__()
        self.layers = list(modules)

    def __call__(self, x):
        for m in self.layers:
            x = m(x)
        return x


class RoPE(Module):
    def __init__(self, dims, traditional=False, base=10000, scale=1.0):
        super().__init__()
        self.dims, self.traditional, self.base, self.scale = dims, traditional, base, scale

    def __call__(self, x, offset=0):
        return mx.fast.rope(x, self.dims, traditional=self.traditional, base=self.base, scale=self.scale, offset=offset)


# ---- quantized layers (mlx.nn.QuantizedLinear / QuantizedEmbedding / nn.quantize, mode="affine"): MLX's documented layer
# semantics over the shim's mx.quantize / mx.quantized_matmul / mx.dequantize (oracle/mlx_shim/mlx/core.py)
class QuantizedLinear(Module):
    def __init__(self, input_dims, output_dims, bias=True, group_size=64, bits=4, mode="affine"):
        super().__init__()
        self.group_size, self.bits, self.mode = group_size, bits, mode
        s = math.sqrt(1.0 / input_dims)
        self.weight, self.scales, self.biases = mx.quantize(_init((output_dims, input_dims), s), group_size, bits, mode=mode)
        if bias:
            self.bias = mx.zeros((output_dims,))

    def _items(self):                       # group_size / bits / mode are attributes, not parameters
        for k, v in self.__dict__.items():
            if not k.startswith("_") and isinstance(v, mx.array):
                yield k, v

    def __call__(self, x):
        x = mx.quantized_matmul(x, self.weight, scales=self.scales, biases=self.biases, transpose=True,
                                group_size=self.group_size, bits=self.bits, mode=self.mode)
        if "bias" in self.__dict__:
            x = x + self.bias
        return x

    @classmethod
    def from_linear(cls, linear_layer, group_size=64, bits=4, mode="affine"):
        output_dims, input_dims = linear_layer.weight.shape
        ql = cls(input_dims, output_dims, False, group_size, bits, mode=mode)
        ql.weight, ql.scales, ql.biases = mx.quantize(linear_layer.weight, group_size, bits, mode=mode)
        if "bias" in linear_layer.__dict__:
            ql.bias = linear_layer.bias
        return ql


class QuantizedEmbedding(Module):
    def __init__(self, num_embeddings, dims, group_size=64, bits=4, mode="affine"):
        super().__init__()
        self.group_size, self.bits, self.mode = group_size, bits, mode
        self.num_embeddings, self.dims = num_embeddings, dims
        self.weight, self.scales, self.biases = mx.quantize(_init((num_embeddings, dims), math.sqrt(1.0 / dims)), group_size, bits, mode=mode)

    _items = QuantizedLinear._items

    def __call__(self, x):
        return mx.dequantize(self.weight[x], scales=self.scales[x], biases=self.biases[x], group_size=self.group_size,
                             bits=self.bits, mode=self.mode)

    def as_linear(self, x):
        return mx.quantized_matmul(x, self.weight, scales=self.scales, biases=self.biases, transpose=True,
                                   group_size=self.group_size, bits=self.bits, mode=self.mode)

    @classmethod
    def from_embedding(cls, embedding_layer, group_size=64, bits=4, mode="affine"):
        n, d = embedding_layer.weight.shape
        ql = cls(n, d, group_size, bits, mode=mode)
        ql.weight, ql.scales, ql.biases = mx.quantize(embedding_layer.weight, group_size, bits, mode=mode)
        return ql


def _linear_to_quantized(self, group_size=64, bits=4, mode="affine", **kw):
    return QuantizedLinear.from_linear(self, group_size, bits, mode=mode)


def _embedding_to_quantized(self, group_size=64, bits=4, mode="affine", **kw):
    return QuantizedEmbedding.from_embedding(self, group_size, bits, mode=mode)


Linear.to_quantized = _linear_to_quantized
Embedding.to_quantized = _embedding_to_quantized


def quantize(model, group_size=None, bits=None, *, mode="affine", quantize_input=False, class_predicate=None):
    """nn.quantize: every leaf module for which class_predicate(path, module) is truthy is replaced by its
    `to_quantized(...)` form (a dict answer carries that module's own parameters); modules without `to_quantized` that
    the predicate accepts are an error, as in MLX."""
    if quantize_input:
        raise NotImplementedError("mlx shim: activation quantization is outside the pinned path")
    class_predicate = class_predicate or (lambda _, m: hasattr(m, "to_quantized"))
    for path, m in model.named_modules():
        if not path or any(isinstance(v, (Module, list, dict)) and not isinstance(v, mx.array) and _has_module(v)
                           for _, v in m._items()):
            continue                                      # not a leaf
        ans = class_predicate(path, m)
        if not ans:
            continue
        if not hasattr(m, "to_quantized"):
            raise ValueError(f"Unable to quantize model of type {type(m)}")
        kwargs = dict(ans) if isinstance(ans, dict) else {"group_size": group_size, "bits": bits, "mode": mode}
        model._set(path, m.to_quantized(**kwargs))
    return model


def _has_module(v):
    if isinstance(v, Module):
        return True
    if isinstance(v, (list, tuple)):
        return any(_has_module(x) for x in v)
    if isinstance(v, dict):
        return any(_has_module(x) for x in v.values())
    return False


class _Placeholder(Module):
    def __init__(self, *a, **k):
        raise NotImplementedError(f"mlx shim: nn.{type(self).__name__} is outside the pinned Qwen2-VL path")


def __getattr__(name):
    if name[:1].isupper():
        return type(name, (_Placeholder,), {})
    raise AttributeError(f"mlx shim (oracle/mlx_shim): mlx.nn.{name} is outside the pinned Qwen2-VL path")
