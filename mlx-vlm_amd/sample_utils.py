"""Sampler front end - same constructor surface as the reference's
mlx_vlm/sample_utils.py:10-89 (make_sampler), but instead of composing Python
closures over mx ops it returns a `Sampler` SPEC that the engine hands to the
fused HIP sampler (csrc/sample.hip: logsumexp -> top-n-sigma -> p-less -> typical-p
-> top-p -> min-p -> xtc -> top-k -> Gumbel-max categorical, the reference's filter order).  The spec is callable
on a logprobs tensor too, for code written against the reference API.
"""
from __future__ import annotations

import itertools
import os
from dataclasses import dataclass
from typing import Optional

import torch

# unseeded samplers: the reference draws from MLX's global RNG state (sample_utils.py:385-387 -> mx.random), so two
# unseeded calls differ.  Here every unseeded sampler takes a fresh seed: process entropy + a counter.
_unseeded = itertools.count(int.from_bytes(os.urandom(4), "little"))


@dataclass
class Sampler:
    temp: float = 0.0
    top_p: float = 0.0
    min_p: float = 0.0
    top_k: int = 0
    seed: int = 0
    min_tokens_to_keep: int = 1
    top_n_sigma: float = 0.0
    p_less: bool = False
    typical_p: float = 1.0
    xtc_probability: float = 0.0
    xtc_threshold: float = 0.0
    xtc_special_tokens: tuple = ()
    _calls: int = 0

    @property
    def greedy(self) -> bool:
        return self.temp == 0

    @property
    def extended(self) -> bool:
        """Filters beyond the three the captured decode step carries in vlm_decode_args (top-p / min-p / top-k): a step with
        one of these runs the forward and then `vlm_sample_ex` on its logits (same kernel, the full parameter block)."""
        return not self.greedy and (self.min_tokens_to_keep != 1 or self.top_n_sigma > 0.0 or bool(self.p_less)
                                    or 0.0 < self.typical_p < 1.0 or self.xtc_probability > 0.0)

    def engine_args(self) -> dict:
        return dict(temperature=float(self.temp), top_p=float(self.top_p if 0 < self.top_p < 1 else 1.0),
                    min_p=float(self.min_p), top_k=int(self.top_k), seed=int(self.seed))

    def sample_args(self) -> dict:
        """keyword arguments of ops.sample (vlm_sample_ex) - python floats, converted on the other side of the C ABI"""
        return dict(self.engine_args(), top_p=float(self.top_p), min_tokens_to_keep=int(self.min_tokens_to_keep),
                    top_n_sigma=float(self.top_n_sigma) if self.top_n_sigma > 0 else 0.0, p_less=bool(self.p_less),
                    typical_p=float(self.typical_p) if 0.0 < self.typical_p < 1.0 else 1.0,
                    xtc_probability=float(self.xtc_probability), xtc_threshold=float(self.xtc_threshold),
                    xtc_special_tokens=list(self.xtc_special_tokens))

    def __call__(self, logprobs: torch.Tensor) -> torch.Tensor:
        """logprobs [B, V] -> tokens [B] (the reference's sampler-closure contract): the same HIP kernel, filtering the
        log-probs as given (greedy: its argmax)."""
        from . import ops

        x = logprobs if logprobs.dim() == 2 else logprobs[None]
        step = torch.tensor([self._calls], dtype=torch.int32, device=x.device)
        self._calls += 1
        tok, _ = ops.sample(x.to(torch.bfloat16).contiguous(), step=step, want_logprobs=False, input_is_logprobs=True,
                            **self.sample_args())
        return tok


def make_sampler(temp: float = 0.0, top_p: float = 0.0, min_p: float = 0.0, min_tokens_to_keep: int = 1,
                 top_k: int = 0, top_n_sigma: float = 0.0, p_less: bool = False, typical_p: float = 1.0,
                 xtc_probability: float = 0.0, xtc_threshold: float = 0.0, xtc_special_tokens=None,
                 seed: Optional[int] = None) -> Sampler:
    """reference sample_utils.py:10-89.  argmax when temp == 0; otherwise the filters in the reference's order (top-n-sigma,
    p-less, typical-p, top-p, min-p, xtc, top-k) then categorical(logprobs / temp) - all inside csrc/sample.hip.  The
    argument checks are the ones the reference's closures make on their first call (160-165, 200-203, 253-260, 323-326,
    363-370), made here at construction."""
    if temp != 0:
        if not (0 <= min_p <= 1.0):
            raise ValueError(f"`min_p` has to be a float in the [0, 1] interval, but is {min_p}")
        if not isinstance(min_tokens_to_keep, int) or min_tokens_to_keep < 1:
            raise ValueError(f"`min_tokens_to_keep` has to be a positive integer, but is {min_tokens_to_keep}")
        # (top_n_sigma <= 0 and typical_p outside (0, 1) never reach their closures in the reference - make_sampler's own
        # conditions, sample_utils.py:69-74 - and are ignored here the same way)
        if xtc_probability > 0.0:
            if not (0 <= xtc_threshold <= 0.5):
                raise ValueError(f"`threshold` has to be a float in the [0, 0.5] interval, but is {xtc_threshold}")
            if not (0 <= xtc_probability <= 1.0):
                raise ValueError(f"`probability` has to be a float in the [0, 1] interval, but is {xtc_probability}")
            if xtc_special_tokens is not None and len(xtc_special_tokens) > 256:
                raise NotImplementedError("more than 256 xtc_special_tokens are not built (csrc/sample.hip)")
        if top_k and (not isinstance(top_k, int) or top_k < 0):
            raise ValueError(f"`top_k` has to be a non-negative integer, but is {top_k}")
    elif not (0 <= min_p <= 1.0):
        raise ValueError(f"`min_p` has to be a float in the [0, 1] interval, but is {min_p}")
    # greedy: the seed is never read, and it is part of the captured step's key - a fresh seed per call would make every
    # default generate() re-capture its decode graph; unseeded SAMPLING draws a fresh stream per sampler
    if seed is None:
        seed = 0 if temp == 0 else next(_unseeded) & 0xFFFFFFFF
    return Sampler(temp=temp, top_p=top_p, min_p=min_p, top_k=top_k, seed=int(seed), min_tokens_to_keep=min_tokens_to_keep,
                   top_n_sigma=top_n_sigma, p_less=bool(p_less), typical_p=typical_p, xtc_probability=xtc_probability,
                   xtc_threshold=xtc_threshold, xtc_special_tokens=tuple(int(t) for t in (xtc_special_tokens or ())))


HIST_CAP = 256      # device token history per decode row (csrc/sample.hip::logit_penalties_kernel)


@dataclass
class LogitsProcessors:
    """What the reference's `make_logits_processors` (sample_utils.py:92-146) builds as a list of Python closures over mx
    ops, as a SPEC for the device pass `vlm_apply_logit_penalties` (same processors, same order, same rounding points):
    logit_bias -> repetition penalty -> presence penalty -> frequency penalty."""
    logit_bias: Optional[dict] = None
    repetition_penalty: float = 0.0
    repetition_context_size: int = 20
    presence_penalty: float = 0.0
    presence_context_size: int = 20
    frequency_penalty: float = 0.0
    frequency_context_size: int = 20

    def __bool__(self):
        return bool(self.logit_bias) or any(p not in (None, 0, 0.0) for p in
                                            (self.repetition_penalty, self.presence_penalty, self.frequency_penalty))

    def key(self):
        return (tuple(sorted((self.logit_bias or {}).items())), self.repetition_penalty, self.repetition_context_size,
                self.presence_penalty, self.presence_context_size, self.frequency_penalty, self.frequency_context_size)


def make_logits_processors(logit_bias=None, repetition_penalty=None, repetition_context_size=20, presence_penalty=None,
                           presence_context_size=20, frequency_penalty=None, frequency_context_size=20) -> LogitsProcessors:
    """reference sample_utils.py:92-146 (argument names, defaults and the repetition-penalty check of 405-406)."""
    if repetition_penalty is not None and (not isinstance(repetition_penalty, (int, float)) or repetition_penalty < 0):
        raise ValueError(f"penalty must be a non-negative float, got {repetition_penalty}")
    for name, c in (("repetition", repetition_context_size), ("presence", presence_context_size),
                    ("frequency", frequency_context_size)):
        if c is not None and int(c) > HIST_CAP:
            raise NotImplementedError(f"{name}_context_size > {HIST_CAP} is not built (device history ring)")
    z = lambda v: 0.0 if v is None else float(v)      # noqa: E731
    n = lambda v: 20 if v is None else max(0, int(v))  # noqa: E731
    return LogitsProcessors(dict(logit_bias) if logit_bias else None, z(repetition_penalty), n(repetition_context_size),
                            z(presence_penalty), n(presence_context_size), z(frequency_penalty), n(frequency_context_size))
