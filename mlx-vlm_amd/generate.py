"""Generation engine with the reference's Python API for the hot path:

    generate_step    reference mlx_vlm/generate/ar.py:151-515
    stream_generate  reference mlx_vlm/generate/dispatch.py:694-1105
    generate         reference mlx_vlm/generate/dispatch.py:1108-1228
    batch_generate   reference mlx_vlm/generate/ar.py:2890-3096 (static batches)
    GenerationResult reference mlx_vlm/generate/common.py:216-240

The device work of a decode step is one hipGraph replay (embedding gather ->
28 x [RMSNorm+QKV GEMV, M-RoPE + paged KV write, split-K paged attention, o_proj
GEMV + residual, RMSNorm + gate/up GEMV + SwiGLU, down GEMV + residual] -> final
norm + lm_head GEMV -> logsumexp/sample -> position advance), all state device
resident.  The host only enqueues replays `lookahead` steps ahead of the token it
is handing to the caller and reads tokens back through pinned memory - the same
double-buffering contract as the reference's mx.async_eval one-step lookahead
(ar.py:498-508), generalised to a configurable depth.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Generator, List, Optional, Tuple, TypedDict, Union

import numpy as np
import torch

from . import _lib, ops
from .models import cache as cache_mod
from .sample_utils import Sampler, make_sampler

DEFAULT_MAX_TOKENS = 2048          # reference generate/common.py:19-24
DEFAULT_TEMPERATURE = 0.0
DEFAULT_TOP_P = 1.0
DEFAULT_TOP_K = 0
DEFAULT_MIN_P = 0.0
DEFAULT_QUANTIZED_KV_START = 5000      # reference generate/common.py:17
DEFAULT_PREFILL_STEP_SIZE = 2048
ONE_SHOT_PREFILL_TOKENS = 32768    # with the default step size, prompts up to this many tokens prefill in one shot (generate_step)


@dataclass
class GenerationResult:
    text: str = ""
    token: Optional[int] = None
    logprobs: Optional[Any] = None
    prompt_tokens: int = 0
    generation_tokens: int = 0
    total_tokens: int = 0
    prompt_tps: float = 0.0
    generation_tps: float = 0.0
    peak_memory: float = 0.0
    cached_tokens: int = 0
    finish_reason: Optional[str] = None


@dataclass
class BatchStats:
    """reference ar.py:863-884"""
    prompt_tokens: int = 0
    prompt_tps: float = 0.0
    prompt_time: float = 0.0
    generation_tokens: int = 0
    generation_tps: float = 0.0
    generation_time: float = 0.0
    peak_memory: float = 0.0


@dataclass
class BatchResponse:
    """reference ar.py:887-902"""
    texts: List[str]
    stats: BatchStats
    tokens: List[List[int]] = field(default_factory=list)
    image_sizes: Optional[List[Tuple[int, int]]] = None


class PromptCacheState:
    """KV cache + token history across conversation turns (reference generate/common.py:243-263)."""

    def __init__(self):
        self.cache: Optional[List[Any]] = None
        self.token_ids: Optional[List[int]] = None

    def find_prefix_length(self, new_ids) -> int:
        if self.token_ids is None:
            return 0
        n = min(len(self.token_ids), len(new_ids))
        for i in range(n):
            if self.token_ids[i] != new_ids[i]:
                return i
        return n

    def update(self, token_ids, kv_cache):
        if self.cache is not None and kv_cache is not self.cache:
            self.release()                      # a replaced cache gives its pool slot and pages back
        self.token_ids = [int(t) for t in token_ids]
        self.cache = kv_cache

    def release(self):
        """Return the pooled KV sequence (slot + pages) of the kept cache to the engine's pool.  The reference's arrays are
        garbage collected; here the pool is a fixed preallocation, so dropping a state without this would leak its slot."""
        cache, self.cache, self.token_ids = self.cache, None, None
        seq = getattr(cache[0], "_seq", None) if cache else None
        if seq is not None and not getattr(seq, "released", False):
            try:
                seq.release()
            except Exception:       # (interpreter teardown: the pool may already be gone)
                pass

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _peak_gb():
    return torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0


class GenerateKwargs(TypedDict, total=False):
    """reference generate/types.py:20-63: the keyword arguments generate / stream_generate / generate_step accept.  The same
    names; array-typed entries are torch tensors or numpy arrays here.  Options outside the built path are accepted by the
    type and REFUSED at the call (NotImplementedError) rather than ignored: kv_key_bits / kv_value_bits / kv_*_scheme
    (TurboQuant), apc_manager / apc_tenant (automatic prefix caching), video."""
    max_tokens: int
    temperature: float
    repetition_penalty: Optional[float]
    repetition_context_size: Optional[int]
    presence_penalty: Optional[float]
    presence_context_size: Optional[int]
    frequency_penalty: Optional[float]
    frequency_context_size: Optional[int]
    top_p: float
    min_p: float
    top_k: int
    logit_bias: Optional[Dict[int, float]]
    prompt_cache: Optional[List[Any]]
    max_kv_size: Optional[int]
    kv_bits: Optional[float]
    kv_key_bits: Optional[float]
    kv_value_bits: Optional[float]
    kv_key_scheme: Optional[str]
    kv_value_scheme: Optional[str]
    kv_group_size: int
    kv_quant_scheme: str
    quantized_kv_start: int
    sampler: Optional[Callable[[Any], Any]]
    logits_processors: Optional[List[Callable[[Any, Any], Any]]]
    prefill_step_size: Optional[int]
    input_ids: Any
    pixel_values: Any
    mask: Any
    resize_shape: Optional[Tuple[int, int]]
    eos_tokens: Optional[List[Any]]
    stopping_criteria: Any
    thinking_budget: Optional[int]
    thinking_end_token: str
    thinking_start_token: Optional[str]
    enable_thinking: bool
    skip_special_tokens: bool
    vision_cache: Any
    prompt_cache_state: Any
    apc_manager: Any
    apc_tenant: Optional[str]
    seed: Optional[int]
    verbose: bool
    video: Any


def _policy_enabled(policy) -> bool:
    return bool(getattr(policy, "enabled", policy))


def _chunked_prefill_enabled(model, *, input_ids=None, inputs_embeds=None, prompt_cache=None, draft_model=None, draft_kind=None,
                             prefill_kwargs=None) -> bool:
    """reference generate/common.py:39-74: may the prompt be fed in `prefill_step_size` chunks?  The model (or its language
    model) decides with a callable `chunked_prefill_policy(...)` -> bool or an object with `.enabled`; a truthy
    `no_chunked_prefill` attribute on either forbids it; otherwise chunking is on unless a draft model is involved."""
    prefill_kwargs = prefill_kwargs or {}
    candidates = [model]
    language_model = getattr(model, "language_model", None)
    if language_model is not None and language_model is not model:
        candidates.append(language_model)
    for candidate in candidates:
        policy = getattr(candidate, "chunked_prefill_policy", None)
        if callable(policy):
            return _policy_enabled(policy(input_ids=input_ids, inputs_embeds=inputs_embeds, prompt_cache=prompt_cache,
                                          draft_model=draft_model, draft_kind=draft_kind, prefill_kwargs=prefill_kwargs))
    if any(getattr(candidate, "no_chunked_prefill", False) for candidate in candidates):
        return False
    return draft_model is None


class _TokenPipe:
    """Pinned-memory token read-back: one slot + event per generated token index."""

    def __init__(self, B: int, cap: int):
        self.buf = torch.empty(cap, B, dtype=torch.int32).pin_memory()
        self.events: List[Optional[torch.cuda.Event]] = [None] * cap
        self.cap = cap

    def push(self, idx: int, src: torch.Tensor):
        self.buf[idx % self.cap].copy_(src, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[idx % self.cap] = ev

    def get(self, idx: int) -> np.ndarray:
        self.events[idx % self.cap].synchronize()
        return self.buf[idx % self.cap].numpy().copy()


def _resolve_sampler(sampler, temperature, top_p, min_p, top_k, seed, **more):
    """-> (device Sampler, python callable or None).  A `sample_utils.Sampler` with top-p / min-p / top-k runs inside the
    captured step; one with the other filters of make_sampler (top_n_sigma, p_less, typical_p, xtc, min_tokens_to_keep - the
    reference's generate_step takes the first three as keywords, ar.py:168-170,279-288) is called on the step's log-probs
    (the same HIP kernel through vlm_sample_ex) around an EAGER step, like any other callable: the reference's
    `sampler(logprobs) -> token` contract (ar.py:151-193,369-379), which gets the log-probs as a device tensor."""
    if sampler is None:
        sampler = make_sampler(temp=temperature, top_p=top_p, min_p=min_p, top_k=top_k, seed=seed,
                               **{k: v for k, v in more.items() if v is not None})
    if isinstance(sampler, Sampler):
        return (make_sampler(temp=0.0), sampler) if sampler.extended else (sampler, None)
    if callable(sampler):
        return make_sampler(temp=0.0), sampler
    raise TypeError("sampler must be a mlx_vlm_amd.sample_utils.Sampler or a callable logprobs -> token")


def _as_token(y) -> int:
    if isinstance(y, torch.Tensor):
        return int(y.reshape(-1)[0].item())
    return int(np.asarray(y).reshape(-1)[0])


def _logprobs_of(logits: torch.Tensor) -> torch.Tensor:
    """logits - logsumexp(logits) in the logits dtype (ar.py:368) by the sampler kernel (the argmax it also returns is unused)"""
    _, lp = ops.sample(logits, want_logprobs=True, temperature=0.0)
    return lp


def generate_step(input_ids, model, pixel_values, mask, *, max_tokens: int = DEFAULT_MAX_TOKENS,
                  temperature: float = DEFAULT_TEMPERATURE, top_p: float = DEFAULT_TOP_P, min_p: float = DEFAULT_MIN_P,
                  top_k: int = DEFAULT_TOP_K, prompt_cache: Optional[List[Any]] = None, sampler=None,
                  logits_processors=None, prefill_step_size: Optional[int] = DEFAULT_PREFILL_STEP_SIZE,
                  seed: Optional[int] = None, lookahead: int = 4, return_logprobs: bool = True, use_graph: bool = True,
                  **kwargs) -> Generator[Tuple[int, Any], None, None]:
    """Yield (token, logprobs) like the reference's generate_step (ar.py:151-515) for ONE sequence.

    input_ids [1, L]; pixel_values / mask as produced by prepare_inputs; extra kwargs
    (image_grid_thw, ...) are forwarded to model.get_input_embeddings.  logprobs is a device
    bf16 [V] tensor (None when return_logprobs=False)."""
    # reference ar.py:151-193,303-304,360-364: `logits_processors` = callables (tokens, logits) -> logits appended to the
    # built-in ones.  A `sample_utils.LogitsProcessors` spec (this package's make_logits_processors) joins the device pass;
    # any other callable forces the EAGER step below (it cannot run inside a captured graph).
    from .sample_utils import LogitsProcessors as _Spec

    py_procs = [p for p in (logits_processors or []) if p is not None and not isinstance(p, _Spec)]
    extra_specs = [p for p in (logits_processors or []) if isinstance(p, _Spec) and p]
    for p in py_procs:
        if not callable(p):
            raise TypeError("logits_processors entries must be callables (tokens, logits) -> logits")
    if len(extra_specs) > 1:
        raise NotImplementedError("one make_logits_processors spec per request")
    from .sample_utils import make_logits_processors

    # reference ar.py:290-302: the penalties / bias of generate_step as a device-side logits pass
    procs = make_logits_processors(kwargs.pop("logit_bias", None), kwargs.pop("repetition_penalty", None),
                                   kwargs.pop("repetition_context_size", 20), kwargs.pop("presence_penalty", None),
                                   kwargs.pop("presence_context_size", 20), kwargs.pop("frequency_penalty", None),
                                   kwargs.pop("frequency_context_size", 20))
    if extra_specs:
        if procs:
            raise NotImplementedError("penalty keyword arguments and a make_logits_processors spec: pass one of them")
        procs = extra_specs[0]
    thinking_budget_criteria = kwargs.pop("thinking_budget_criteria", None)
    if kwargs.pop("draft_model", None):
        # the reference would switch the decoding scheme (speculative): dropping the request silently would change results
        # without telling the caller
        raise NotImplementedError("draft_model is outside the built hot path (SURVEY section 8f.4)")
    # reference ar.py:173,310-315: make_prompt_cache(model.language_model, max_kv_size) - a RotatingKVCache(max_kv_size, keep=4)
    # per layer when the caller brings no prompt_cache.  Here: a bound on the paged sequence, served by eager steps
    max_kv_size = kwargs.pop("max_kv_size", None)
    # uniform quantized KV cache (reference ar.py:174-181,249-260,362: maybe_quantize_kv_cache after EVERY forward): from the
    # first forward that leaves the cache at quantized_kv_start tokens or more, the cache is a QuantizedKVCache
    kv_bits = kwargs.pop("kv_bits", None)
    kv_group_size = kwargs.pop("kv_group_size", None) or 64
    quantized_kv_start = kwargs.pop("quantized_kv_start", None)
    quantized_kv_start = DEFAULT_QUANTIZED_KV_START if quantized_kv_start is None else int(quantized_kv_start)
    kv_scheme = kwargs.pop("kv_quant_scheme", None)
    if kv_bits is not None:
        if kv_scheme not in (None, "uniform") or float(kv_bits) != 8 or int(kv_group_size) != 64:
            raise NotImplementedError(f"kv_bits={kv_bits} kv_group_size={kv_group_size} kv_quant_scheme={kv_scheme}: the uniform "
                                      "8-bit / group-64 quantized KV cache is built (TurboQuant / other widths: SURVEY section 2 out of scope)")
    for k in ("verbose", "prompt_cache_checkpoint", "prompt_cache_checkpoint_len"):
        kwargs.pop(k, None)
    smp, py_sampler = _resolve_sampler(sampler, temperature, top_p, min_p, top_k, seed, top_n_sigma=kwargs.pop("top_n_sigma", None),
                                       p_less=kwargs.pop("p_less", None), typical_p=kwargs.pop("typical_p", None))
    sargs = smp.engine_args()
    if isinstance(py_sampler, Sampler):
        # an extended Sampler runs through its __call__, whose RNG step is a counter of the object: like the captured path
        # (step 0 at the start of every generation) a seeded sampler reused across two generations draws the same stream
        py_sampler._calls = 0
    eager = bool(py_procs) or py_sampler is not None or thinking_budget_criteria is not None or max_kv_size is not None
    lm = model.language_model

    ids = input_ids.detach().cpu().numpy() if isinstance(input_ids, torch.Tensor) else np.asarray(input_ids)
    if ids.ndim == 1:
        ids = ids[None]
    if ids.shape[0] != 1:
        raise ValueError("generate_step handles one sequence; use batch_generate for batches")

    # continuation of a cached prefix (stream_generate's prompt_cache_state path): `ids` is the uncached suffix, its rope
    # positions / delta come from the FULL prompt (reference dispatch.py:612-649 primes them on the model object)
    pos_override, delta_override = kwargs.pop("position_ids", None), kwargs.pop("rope_deltas", None)
    f = model.get_input_embeddings(ids, pixel_values, mask=mask, **kwargs)
    if pos_override is not None:
        f.position_ids = np.asarray(pos_override)
    if delta_override is not None:
        f.rope_deltas = np.asarray(delta_override)
    own_cache = prompt_cache is None
    if own_cache:
        prompt_cache = cache_mod.make_prompt_cache(lm, max_kv_size=max_kv_size)
    if prompt_cache[0]._seq.rotating:
        eager = True                                   # (also a caller's own make_prompt_cache(lm, max_kv_size=...))
        if kv_bits is not None:
            raise NotImplementedError("RotatingKVCache Quantization NYI")          # the reference's words (cache.py:583-584)
        # the reference prefills prompts beyond prefill_step_size in chunks (ar.py:425-470) and a rotating cache trims its
        # window between chunks (cache.py:486-505): only the single-chunk prompt is built
        if prefill_step_size is not None and ids.shape[1] > int(prefill_step_size):
            raise NotImplementedError(f"max_kv_size with a prompt of {ids.shape[1]} tokens > prefill_step_size {prefill_step_size}: "
                                      "the reference feeds such a prompt in chunks and its rotating cache re-orders / trims the "
                                      "window between them; pass a larger prefill_step_size (one chunk) ")
    emb = f.inputs_embeds
    L = emb.shape[1]
    pos = np.asarray(f.position_ids)
    if pos.ndim == 2:
        pos = np.broadcast_to(pos[None], (3,) + pos.shape)

    def maybe_quantize_kv_cache():      # generate/common.py:170-181 on the paged cache: all layers share one offset
        sq = prompt_cache[0]._seq
        if kv_bits is not None and not sq.q8 and sq.offset >= quantized_kv_start:
            lm.quantize_kv([sq], bits=int(kv_bits), group_size=int(kv_group_size))

    # chunked prefill (reference ar.py:409-470): a prompt longer than prefill_step_size is fed in chunks of that many tokens -
    # every chunk attends to the cached tokens and itself (LanguageModel._prefill_onto_cache) - until ONE token is left, which
    # the final call turns into the first logits.  The model may forbid it (generate/common.py:39-74: chunked_prefill_policy /
    # no_chunked_prefill); a rotating window takes its first prompt whole (refused above).
    E, P = emb.reshape(L, -1), pos.reshape(3, L)
    step_size = None if prefill_step_size is None else int(prefill_step_size)
    if step_size is not None and not _chunked_prefill_enabled(model, input_ids=ids, inputs_embeds=emb, prompt_cache=prompt_cache,
                                                              draft_model=None, draft_kind=None, prefill_kwargs=kwargs):
        step_size = None
    # The reference chunks to bound MLX's prompt memory; here the one-shot prefill kernels take any prompt the workspaces hold
    # and the chunk path re-attends over the gathered prefix (LanguageModel._prefill_onto_cache): measured on the 2B, first token
    # of a 4096 / 8192 / 16384-token prompt 37 / 88 / 276 ms in chunks of 2048 against 22 / 38 / 90 ms in one shot
    # (scripts/r06/long_prompt.py).  So the DEFAULT step size means "the engine decides": prompts up to
    # ONE_SHOT_PREFILL_TOKENS go in one shot (same logits up to summation order); any other value is honoured as given.
    if step_size == DEFAULT_PREFILL_STEP_SIZE and L <= ONE_SHOT_PREFILL_TOKENS:
        step_size = None
    done = 0
    if step_size is not None and step_size > 0 and L > step_size and not prompt_cache[0]._seq.rotating:
        while L - done > 1:
            n = min(step_size, L - done - 1)
            lm.prefill(E[done:done + n], P[:, done:done + n], [prompt_cache], [n], "last", reserve_extra=L - done - n + max_tokens + 2)
            # (the reference quantises between chunks, common.py:170-181; here the cache turns 8-bit AFTER the last prefill call:
            #  the chunk path attends over the bf16 pages - LanguageModel._prefill_onto_cache - and the whole-prompt path always
            #  quantised behind the prefill.  The chunks after quantized_kv_start so see unrounded K / V: a deviation of the
            #  PROMPT pass only, the decode steps attend over the same 8-bit pools either way)
            done += n
    logits = lm.prefill(E[done:], P[:, done:], [prompt_cache], [L - done], "last", reserve_extra=max_tokens + 2)

    if eager:
        yield from _generate_step_eager(lm, ids, logits, prompt_cache, f, procs, py_procs, smp, sargs, py_sampler,
                                        thinking_budget_criteria, max_tokens, return_logprobs, maybe_quantize_kv_cache, own_cache)
        return
    maybe_quantize_kv_cache()
    step0 = torch.zeros(1, dtype=torch.int32, device=logits.device)
    if procs:
        # ar.py:360-364: at the first step `tokens` is the prompt; the history then grows by every token fed back (the
        # decode step pushes its input token before it applies the processors)
        pst = lm.decode_state(1)
        pst.set_history([ids.reshape(-1)])
        ops.apply_logit_penalties(logits, pst.penalty_args(procs))
    tok0, lp0 = ops.sample(logits, step=step0, want_logprobs=return_logprobs, **sargs)

    deltas = np.asarray(f.rope_deltas).reshape(-1)[:1]
    st = lm.decode_begin([prompt_cache], tok0, deltas, max_new_tokens=max_tokens + 1)
    st.step.fill_(1)  # sampling step counter: token 0 used step 0
    seq = prompt_cache[0]._seq
    base_offset = seq.offset

    lookahead = max(1, min(int(lookahead), st.ring_len - 1))
    pipe = _TokenPipe(1, cap=lookahead + 2)
    pipe.push(0, tok0)
    lps: List[Optional[torch.Tensor]] = [lp0[0] if lp0 is not None else None] + [None] * (lookahead + 1)
    issued = 0   # decode steps enqueued; step s (0-based) produces generated token s + 1
    n = 0        # tokens handed to the caller (counted BEFORE the yield: a consumer that closes the generator at the
                 # yield - the stop-token path of stream_generate - has received that token)
    try:
        while n < max_tokens:
            while issued < min(n + lookahead, max_tokens - 1):
                maybe_quantize_kv_cache()          # (the forward just enqueued may have carried the cache past the start)
                lm.decode_run(st, 1, sargs, use_graph=use_graph, penalties=procs if procs else None)
                issued += 1
                pipe.push(issued, st.tok)
                if return_logprobs:
                    lps[issued % (lookahead + 2)] = st.logprobs[0].clone()
            tok = int(pipe.get(n)[0])
            n += 1
            yield tok, lps[(n - 1) % (lookahead + 2)]
    finally:
        # the cache holds prompt + the tokens that were FED back: every yielded token except the last (steps enqueued
        # ahead of the caller wrote further slots; they are overwritten when decoding continues from this offset)
        seq.offset = base_offset + max(0, min(n, issued + 1) - 1)
        if own_cache:
            seq.release()


def _generate_step_eager(lm, ids, logits, prompt_cache, f, procs, py_procs, smp, sargs, py_sampler, budget, max_tokens,
                         return_logprobs, maybe_quantize_kv_cache, own_cache):
    """The reference's `_step` loop (ar.py:334-389) run EAGERLY, one forward per token through the module contract
    (`language_model(y, cache=...)` = vlm_llm_decode_forward, logits only): the route for what cannot live inside a captured
    step - Python `logits_processors` callables (tokens, logits) -> logits, a Python `sampler(logprobs) -> token`, and the
    thinking budget (utils.py:2252-2335, ar.py:369-379: the criteria may force the token).  Callables receive DEVICE tensors:
    `tokens` int32 [n] = the prompt followed by every token fed back (ar.py:360), `logits` bf16 [1, V]; they must return a
    [1, V] tensor.  Order as in the reference: built-in processors, then the caller's, then the cache switch-over, then
    log-probs, then the sampler."""
    seq = prompt_cache[0]._seq
    dev = logits.device
    lm._rope_deltas = np.asarray(f.rope_deltas).reshape(-1, 1)[:1]
    need_tokens = bool(py_procs)
    tokens = _lib_h2d(np.asarray(ids, dtype=np.int32).reshape(-1), dev) if need_tokens else None
    pst = None
    if procs:
        pst = lm.decode_state(1)
        pst.set_history([np.asarray(ids).reshape(-1)])
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    fed: Optional[torch.Tensor] = None          # the token fed by the forward that produced `logits` (None: the prompt)
    n = 0
    try:
        while n < max_tokens:
            logits = logits.reshape(1, -1)
            if procs:
                ops.apply_logit_penalties(logits, pst.penalty_args(procs), push_tok=fed)
            for proc in py_procs:
                out = proc(tokens, logits)
                if not isinstance(out, torch.Tensor):
                    out = torch.as_tensor(np.asarray(out), device=dev)
                logits = out.to(device=dev, dtype=logits.dtype).reshape(1, -1).contiguous()
            maybe_quantize_kv_cache()
            if py_sampler is not None:
                lp = _logprobs_of(logits)
                y = _as_token(py_sampler(lp))
            else:
                tok, lp = ops.sample(logits, step=step, want_logprobs=return_logprobs or budget is not None, **sargs)
                y = int(tok.reshape(-1)[0].item())
            if budget is not None and n > 0:
                # ar.py:510-513: after the caller has seen the previous token (it calls criteria(token), dispatch.py:1016-1018)
                # the criteria may have a forced token pending - it REPLACES the sampled one; the log-probs stay the model's
                forced = budget.pop_forced_token_id()
                if forced is not None:
                    y = int(forced)
            step += 1
            n += 1
            yield y, (lp[0] if (lp is not None and return_logprobs) else None)
            if n >= max_tokens:
                break
            fed = _lib_h2d(np.asarray([y], dtype=np.int32), dev)
            if need_tokens:
                tokens = torch.cat([tokens, fed])
            logits = lm(np.array([[y]], dtype=np.int64), cache=prompt_cache).logits[:, -1, :]
    finally:
        if own_cache:
            seq.release()


def _lib_h2d(a, dev):
    from ._lib import h2d

    return h2d(a, dev)


# ---------------------------------------------------------------------------------------------
def _tokenizer_of(processor):
    return processor.tokenizer if hasattr(processor, "tokenizer") else processor


def stream_generate(model, processor, prompt: Optional[str] = None, image=None, audio=None, video=None,
                    **kwargs) -> Generator[GenerationResult, None, None]:
    """reference dispatch.py:694-1105 (no APC / vision-cache / prefix-cache reuse: SURVEY §2.1 out of scope).
    Timing semantics kept: prompt_tps = prompt tokens / wall time to the first token (ViT + projector + LLM
    prefill + first sample); generation_tps = tokens / wall time since the first token."""
    from .utils import make_streaming_detokenizer, prepare_inputs

    tokenizer = _tokenizer_of(processor) if processor is not None else None
    kwargs.pop("verbose", None)
    skip_special_tokens = kwargs.pop("skip_special_tokens", False)
    skip_ids = set(tokenizer.all_special_ids) if (skip_special_tokens and hasattr(tokenizer, "all_special_ids")) else set()
    vision_cache = kwargs.pop("vision_cache", None)
    prompt_cache_state = kwargs.pop("prompt_cache_state", None)
    thinking_budget = kwargs.pop("thinking_budget", None)
    thinking_end_token = kwargs.pop("thinking_end_token", "</think>")
    thinking_start_token = kwargs.pop("thinking_start_token", "<think>")
    enable_thinking = kwargs.pop("enable_thinking", False)
    for k in ("resize_shape", "apc_manager", "apc_tenant", "eos_tokens", "stopping_criteria"):
        kwargs.pop(k, None)
    if audio or video:
        raise NotImplementedError("audio / video inputs are outside the built hot path")

    if kwargs.get("input_ids", None) is not None:          # pre-tokenised bypass, dispatch.py:759-762
        input_ids = kwargs.pop("input_ids")
        pixel_values = kwargs.pop("pixel_values", None)
        mask = kwargs.pop("mask", None)
    else:
        inputs = prepare_inputs(processor, images=image, prompts=prompt)
        input_ids = inputs["input_ids"]
        pixel_values = inputs.get("pixel_values")
        mask = inputs.get("attention_mask")
        kwargs.update({k: v for k, v in inputs.items() if k not in ("input_ids", "pixel_values", "attention_mask")})

    ids = np.asarray(input_ids if not isinstance(input_ids, torch.Tensor) else input_ids.cpu())
    total_prompt_tokens = int(ids.size)
    full_ids = [int(t) for t in ids.reshape(-1)]

    # vision feature cache (reference dispatch.py:800-809): the projected features of an image seen before are reused,
    # the ViT prefill of this turn is skipped
    if vision_cache is not None and image is not None and pixel_values is not None and hasattr(model, "encode_image"):
        feats = vision_cache.get(image)
        if feats is None:
            feats = model.encode_image(pixel_values, **{k: v for k, v in kwargs.items()
                                                        if k in ("image_grid_thw", "image_sizes", "pixel_attention_mask")})
            vision_cache.put(image, feats)
        kwargs["cached_image_features"] = feats

    # prompt cache reuse across turns (reference dispatch.py:861-882): the KV cache of the previous turn is trimmed to
    # the shared token prefix and only the suffix is prefilled ONTO it (LanguageModel._prefill_onto_cache)
    cached_tokens = 0
    if prompt_cache_state is not None and prompt_cache_state.cache is not None and ids.ndim == 2 and ids.shape[0] == 1:
        kv = prompt_cache_state.cache
        have = int(kv[0].offset)
        prefix = min(prompt_cache_state.find_prefix_length(full_ids), have)       # never beyond what the cache holds
        cfgm = model.config
        media = {getattr(cfgm, n, None) for n in ("image_token_id", "video_token_id", "image_token_index")} - {None}
        suffix_text_only = not any(t in media for t in full_ids[prefix:])
        lm = model.language_model
        # (a rotating window - max_kv_size - is not continued: the reference reuses one only until it wraps, dispatch.py:656-690,
        # and a multi-token update of it re-orders and trims the window; the turn is prefilled cold)
        if 0 < prefix < len(full_ids) and suffix_text_only and hasattr(lm, "get_rope_index") and not kv[0]._seq.rotating:
            pos, deltas = lm.get_rope_index(ids, kwargs.get("image_grid_thw"), kwargs.get("video_grid_thw"), None)
            pos = np.asarray(pos)
            if pos.ndim == 2:
                pos = np.broadcast_to(pos[None], (3,) + pos.shape)
            for c in kv:
                c.trim(have - prefix)
            kwargs.update(prompt_cache=kv, position_ids=pos[..., prefix:], rope_deltas=deltas)
            ids, pixel_values, cached_tokens = ids[:, prefix:], None, prefix
            kwargs.pop("cached_image_features", None)
    if prompt_cache_state is not None and "prompt_cache" not in kwargs:
        from .models import cache as _cm

        if prompt_cache_state.cache is not None:
            prompt_cache_state.cache[0]._seq.release()        # cold prefill: the old turn's pages go back to the pool
            prompt_cache_state.cache = None
        kwargs["prompt_cache"] = _cm.make_prompt_cache(model.language_model, max_kv_size=kwargs.get("max_kv_size"))
    detok = make_streaming_detokenizer(processor) if processor is not None else None
    stop = getattr(tokenizer, "stopping_criteria", None) if tokenizer is not None else None
    # thinking budget (reference dispatch.py:930-947,1016-1018): the criteria watches every token; past the budget inside a
    # thinking block it forces "\n</think>" through generate_step (which then steps eagerly)
    thinking_criteria = None
    if thinking_budget is not None and tokenizer is not None:
        from .utils import ThinkingBudgetCriteria

        start_id = tokenizer.encode(thinking_start_token, add_special_tokens=False)[-1]
        thinking_criteria = ThinkingBudgetCriteria(tokenizer=tokenizer, thinking_budget=thinking_budget,
                                                   thinking_end_token=thinking_end_token, thinking_start_token=thinking_start_token,
                                                   enable_thinking=enable_thinking,
                                                   prompt_preopens_thinking=start_id in full_ids)
        kwargs["thinking_budget_criteria"] = thinking_criteria
    if tokenizer is not None:
        try:
            tokenizer.thinking_budget_criteria = thinking_criteria
        except Exception:
            pass

    gen = generate_step(ids, model, pixel_values, mask, **kwargs)
    tic = time.perf_counter()
    finish_reason = None
    token, logprobs, n = None, None, -1
    prompt_tps = 0.0
    fed: List[int] = []                 # tokens whose K/V the cache holds after the run (all yielded but the last)
    for n, (token, logprobs) in enumerate(gen):
        fed.append(int(token))
        if n == 0:
            prompt_time = time.perf_counter() - tic
            prompt_tps = total_prompt_tokens / prompt_time
            tic = time.perf_counter()
        if thinking_criteria is not None:
            thinking_criteria(token)
        if stop is not None and stop(token):
            finish_reason = "stop"
            break
        if detok is not None:
            detok.add_token(token, skip_special_token_ids=skip_ids)
        yield GenerationResult(text=detok.last_segment if detok else "", token=token, logprobs=logprobs,
                               prompt_tokens=total_prompt_tokens, generation_tokens=n + 1,
                               total_tokens=total_prompt_tokens + n + 1, prompt_tps=prompt_tps,
                               generation_tps=(n + 1) / max(time.perf_counter() - tic, 1e-9), peak_memory=_peak_gb(),
                               cached_tokens=cached_tokens)
    else:
        finish_reason = "length"
    gen.close()
    if prompt_cache_state is not None:
        kv = kwargs["prompt_cache"]
        # the cache holds the prompt + every generated token that was fed back (generate_step leaves the offset there)
        prompt_cache_state.update((full_ids + fed)[: int(kv[0].offset)], kv)
    if n < 0:
        yield GenerationResult(prompt_tokens=total_prompt_tokens, total_tokens=total_prompt_tokens,
                               peak_memory=_peak_gb(), finish_reason="length")
        return
    if detok is not None:
        detok.finalize()
    yield GenerationResult(text=detok.last_segment if detok else "", token=token, logprobs=logprobs,
                           prompt_tokens=total_prompt_tokens, generation_tokens=n + 1,
                           total_tokens=total_prompt_tokens + n + 1, prompt_tps=prompt_tps,
                           generation_tps=(n + 1) / max(time.perf_counter() - tic, 1e-9), peak_memory=_peak_gb(),
                           cached_tokens=cached_tokens, finish_reason=finish_reason)


def generate(model, processor, prompt: Optional[str] = None, image=None, audio=None, video=None, verbose: bool = False,
             **kwargs) -> GenerationResult:
    """reference dispatch.py:1108-1228."""
    tokenizer = _tokenizer_of(processor) if processor is not None else None
    eos_tokens = kwargs.get("eos_tokens", None)
    stopping_criteria = kwargs.get("stopping_criteria", None)
    if tokenizer is not None and hasattr(tokenizer, "stopping_criteria"):
        if eos_tokens is not None:
            tokenizer.stopping_criteria.add_eos_token_ids(eos_tokens)
        elif stopping_criteria is not None:
            if not callable(stopping_criteria):
                raise ValueError("stopping_criteria must be an instance of StoppingCriteria or a callable")
            tokenizer.stopping_criteria = stopping_criteria
        else:
            tokenizer.stopping_criteria.reset(model.config.eos_token_id)
    text, last = "", None
    for r in stream_generate(model, processor, prompt, image, audio, video, **kwargs):
        if verbose:
            print(r.text, end="", flush=True)
        text += r.text
        last = r
    if last is None:
        return GenerationResult(text=text, peak_memory=_peak_gb())
    if verbose:
        print("\n" + "=" * 10)
        print(f"Prompt: {last.prompt_tokens} tokens, {last.prompt_tps:.3f} tokens-per-sec")
        print(f"Generation: {last.generation_tokens} tokens, {last.generation_tps:.3f} tokens-per-sec")
        print(f"Peak memory: {last.peak_memory:.3f} GB")
    return GenerationResult(text=text, token=last.token, logprobs=last.logprobs, prompt_tokens=last.prompt_tokens,
                            generation_tokens=last.generation_tokens, total_tokens=last.total_tokens,
                            prompt_tps=last.prompt_tps, generation_tps=last.generation_tps, peak_memory=last.peak_memory,
                            cached_tokens=last.cached_tokens, finish_reason=last.finish_reason)


# ---------------------------------------------------------------------------------------------
def _embed_requests_per_model(model, input_ids_list, pixel_values_list, extras):
    """Model families other than Qwen2-VL (llava_bunny, phi3_v): every request through the model's own
    `get_input_embeddings` (its tower, projector and splice rule), the results concatenated for one varlen prefill."""
    embs, poss, lens, deltas = [], [], [], []
    # one tower pass for the images of ALL requests where the model offers it (`encode_images_batched`: per-request features,
    # computed together - the GEMMs of 4 x 729 rows of one request run at a fraction of the rate of 32 x 729)
    feats = None
    if hasattr(model, "encode_images_batched") and sum(p is not None for p in pixel_values_list) > 1:
        feats = model.encode_images_batched(pixel_values_list, extras)
    for j, (ids, pix, kw) in enumerate(zip(input_ids_list, pixel_values_list, extras)):
        ids = np.asarray(ids).reshape(1, -1)
        kw = dict(kw or {})
        if feats is not None and feats[j] is not None:
            kw["cached_image_features"] = feats[j]
        f = model.get_input_embeddings(ids, torch.as_tensor(pix) if pix is not None else None, **kw)
        L = f.inputs_embeds.shape[1]
        p = np.asarray(f.position_ids)
        if p.ndim == 2:
            p = np.broadcast_to(p[None], (3,) + p.shape)
        embs.append(f.inputs_embeds.reshape(L, -1))
        poss.append(p.reshape(3, L))
        lens.append(L)
        deltas.append(int(np.asarray(f.rope_deltas).reshape(-1)[0]) if f.rope_deltas is not None else 0)
    return torch.cat(embs, dim=0), np.concatenate(poss, axis=1), lens, deltas


def embed_requests(model, input_ids_list, pixel_values_list, grids, extras=None):
    """Input embeddings of several requests for ONE varlen prefill: one ViT call over the concatenated patches of all
    images (as the reference does per shape group, ar.py:3165-3167), features scattered into each request's
    placeholder rows, per-request M-RoPE positions.  -> (embeds [sum L, D], position_ids [3, sum L], lengths, rope deltas)
    `extras`: per-request keyword arguments of `get_input_embeddings` besides the pixels (phi3_v: image_sizes)."""
    if not hasattr(model, "merge_input_ids_with_image_features"):
        return _embed_requests_per_model(model, input_ids_list, pixel_values_list, extras or [None] * len(input_ids_list))
    lm = model.language_model
    embs, poss, lens, deltas = [], [], [], []
    has_pix = [p is not None for p in pixel_values_list]
    feats_all = None
    if any(has_pix):
        grid_all = np.concatenate([np.asarray(g) for g, h in zip(grids, has_pix) if h], axis=0)
        pv = _lib.h2d_cat([p for p, h in zip(pixel_values_list, has_pix) if h], lm.device)
        feats_all = model.vision_tower(pv, grid_all)
    foff = 0
    mm = model.config.vision_config.spatial_merge_size ** 2
    for ids, pix, grid in zip(input_ids_list, pixel_values_list, grids):
        ids = np.asarray(ids).reshape(1, -1)
        emb = lm.embed_tokens(ids)
        if pix is not None:
            nfeat = int(np.prod(np.asarray(grid), axis=1).sum()) // mm
            emb = model.merge_input_ids_with_image_features(model.config.image_token_id, model.config.video_token_id,
                                                            feats_all[foff:foff + nfeat], emb, ids)
            foff += nfeat
            p, d = lm.get_rope_index(ids, np.asarray(grid), None, None)
        else:
            p, d = lm.get_rope_index(ids)
            p = np.broadcast_to(p[None], (3,) + p.shape)
        embs.append(emb.reshape(ids.shape[1], -1))
        poss.append(np.asarray(p).reshape(3, -1))
        lens.append(ids.shape[1])
        deltas.append(int(np.asarray(d).reshape(-1)[0]))
    return torch.cat(embs, dim=0), np.concatenate(poss, axis=1), lens, deltas


def batch_generate_ids(model, input_ids_list: List[np.ndarray], pixel_values_list: List[Any], grids: List[Any], *,
                       max_tokens: int = 128, stop_ids=(), sampler: Optional[Sampler] = None, lookahead: int = 4,
                       use_graph: bool = True, extras: Optional[List[Optional[dict]]] = None) -> Tuple[List[List[int]], BatchStats]:
    """Pre-tokenised batched generation: the requests are processed in decode batches of up to 16 sequences.
    One ViT call over the concatenated patches of the batch (as the reference does per shape group,
    ar.py:3165-3167), one varlen LLM prefill, then batched graph decode.  -> (tokens per request, stats).
    `extras`: per-request keyword arguments of the family's `get_input_embeddings` (phi3_v: image_sizes; idefics2:
    pixel_attention_mask), as `generate_batch_continuous` takes them."""
    lm = model.language_model
    smp = sampler or make_sampler()
    if smp.extended:
        raise NotImplementedError("static batches sample inside the captured step (top-p / min-p / top-k); the other filters of "
                                  "make_sampler run through the continuous generator")
    sargs = smp.engine_args()
    stats = BatchStats()
    outs: List[List[int]] = [[] for _ in input_ids_list]
    stop_ids = set(stop_ids)
    order = list(range(len(input_ids_list)))
    i = 0
    while i < len(order):
        rem = len(order) - i
        wide = rem >= 16 and len(lm.pool._free_seqs) >= 16      # 16-row steps: gemv_mfma.hip
        B = 16 if wide else 8 if rem >= 8 else 4 if rem >= 4 else 2 if rem >= 2 else 1
        idxs = order[i:i + B]
        i += B
        t0 = time.perf_counter()
        emb_cat, pos_cat, lens, deltas = embed_requests(model, [input_ids_list[j] for j in idxs],
                                                        [pixel_values_list[j] for j in idxs], [grids[j] for j in idxs],
                                                        extras=[extras[j] for j in idxs] if extras else None)
        caches = lm.make_cache_batch(len(idxs))
        logits = lm.prefill(emb_cat, pos_cat, caches, lens, "last", reserve_extra=max_tokens + 2)
        step0 = torch.zeros(1, dtype=torch.int32, device=logits.device)
        tok0, _ = ops.sample(logits, step=step0, want_logprobs=False, **sargs)
        st = lm.decode_begin(caches, tok0, deltas, max_new_tokens=max_tokens + 1)
        st.step.fill_(1)
        pipe = _TokenPipe(B, cap=lookahead + 2)
        pipe.push(0, tok0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        stats.prompt_tokens += int(sum(lens))
        stats.prompt_time += t1 - t0
        done = [False] * B
        issued, n = 0, 0
        while n < max_tokens and not all(done):
            while issued < min(n + lookahead, max_tokens - 1):
                lm.decode_run(st, 1, sargs, use_graph=use_graph)
                issued += 1
                pipe.push(issued, st.tok)
            toks = pipe.get(n)
            for b, j in enumerate(idxs):
                if done[b]:
                    continue
                t = int(toks[b])
                if t in stop_ids:
                    done[b] = True
                    continue
                outs[j].append(t)
                stats.generation_tokens += 1
            n += 1
        torch.cuda.synchronize()
        stats.generation_time += time.perf_counter() - t1
        for c in caches:
            c[0]._seq.release()
    stats.prompt_tps = stats.prompt_tokens / max(stats.prompt_time, 1e-9)
    stats.generation_tps = stats.generation_tokens / max(stats.generation_time, 1e-9)
    stats.peak_memory = _peak_gb()
    return outs, stats


def batch_generate(model, processor, images=None, audios=None, prompts: Optional[List[str]] = None, max_tokens: int = 128,
                   verbose: bool = False, group_by_shape: bool = True, track_image_sizes: bool = True, **kwargs) -> BatchResponse:
    """reference ar.py:2890-3096: prompts (+ one image each) -> BatchResponse.  Requests are tokenised on the host and
    queued on a `BatchGenerator` (continuous batching, as the reference's `_generate_batch` does, ar.py:3199-3232);
    `continuous=False` runs them as static decode batches of up to 16 (`batch_generate_ids`).
    `group_by_shape` (ar.py:2972-2986, utils.py:2139-2188) is accepted and has nothing to do here: the reference groups
    images of equal size because its vision tower takes one shape per call; this engine's ViT attention is varlen
    (`cu_seqlens`), so images of ANY sizes share one launch and no padding or regrouping exists.  `track_image_sizes`
    fills `BatchResponse.image_sizes` with each image's original (height, width), (0, 0) for text-only prompts."""
    from .batch import generate_batch_continuous
    from .utils import prepare_inputs

    if audios:
        raise NotImplementedError("audio inputs are outside the built hot path")
    prompts = list(prompts or [])
    images = list(images) if images is not None else [None] * len(prompts)
    tokenizer = _tokenizer_of(processor)
    ids_l, pix_l, grid_l, sizes, extras = [], [], [], [], []
    for p, im in zip(prompts, images):
        sizes.append(_image_hw(im))
        inp = prepare_inputs(processor, images=im, prompts=p)
        ids_l.append(np.asarray(inp["input_ids"]).reshape(-1))
        pix_l.append(inp.get("pixel_values"))
        grid_l.append(inp.get("image_grid_thw"))
        # the model family's own get_input_embeddings arguments (phi3_v: image_sizes, idefics2: pixel_attention_mask)
        extras.append({k: v for k, v in inp.items() if k not in ("input_ids", "pixel_values", "image_grid_thw", "attention_mask")})
    stop = getattr(tokenizer, "stopping_criteria", None)
    stop_ids = tuple(getattr(stop, "eos_token_ids", ()) or ())
    smp, py_smp = _resolve_sampler(kwargs.pop("sampler", None), kwargs.pop("temperature", 0.0), kwargs.pop("top_p", 1.0),
                                   kwargs.pop("min_p", 0.0), kwargs.pop("top_k", 0), kwargs.pop("seed", None),
                                   top_n_sigma=kwargs.pop("top_n_sigma", None), p_less=kwargs.pop("p_less", None),
                                   typical_p=kwargs.pop("typical_p", None))
    if py_smp is not None:
        smp = py_smp               # a Python callable: the generator runs eager steps around it (batch.py)
    # the reference's batch_generate hands its penalty keywords to the generator (ar.py:2890-3096): one spec for every request
    from .sample_utils import make_logits_processors
    procs = make_logits_processors(kwargs.pop("logit_bias", None), kwargs.pop("repetition_penalty", None),
                                   kwargs.pop("repetition_context_size", 20), kwargs.pop("presence_penalty", None),
                                   kwargs.pop("presence_context_size", 20), kwargs.pop("frequency_penalty", None),
                                   kwargs.pop("frequency_context_size", 20))
    # decode rows of the generator (the reference's completion_batch_size, ar.py:2584-2606): 16 by default; more (up to 64) run as
    # WIDE steps where the engine and the KV pool's sequence slots (2 * rows + 2) allow
    rows = kwargs.pop("completion_batch_size", None) or kwargs.pop("batch_size", None)
    if kwargs.pop("continuous", True):
        toks, stats = generate_batch_continuous(model, ids_l, pix_l, grid_l, max_tokens=max_tokens, stop_ids=stop_ids, sampler=smp,
                                                extras=extras, logits_processors=procs if procs else None,
                                                **({"batch_size": int(rows)} if rows else {}))
    else:
        if procs or py_smp is not None:
            raise NotImplementedError("static batches (continuous=False) run without logits processors / Python samplers; "
                                      "use the continuous generator")
        toks, stats = batch_generate_ids(model, ids_l, pix_l, grid_l, max_tokens=max_tokens, stop_ids=stop_ids, sampler=smp,
                                         extras=extras)
    texts = [tokenizer.decode(t) if hasattr(tokenizer, "decode") else "" for t in toks]
    return BatchResponse(texts=texts, stats=stats, tokens=toks, image_sizes=sizes if track_image_sizes else None)


def _image_hw(im) -> Tuple[int, int]:
    """original (height, width) of one request's image (reference ar.py:2957-2970): PIL image, array (H, W, C) or path"""
    if im is None:
        return (0, 0)
    if isinstance(im, (list, tuple)):
        return _image_hw(im[0]) if im else (0, 0)
    if hasattr(im, "height") and hasattr(im, "width"):
        return (int(im.height), int(im.width))
    if hasattr(im, "shape") and len(im.shape) >= 2:
        return (int(im.shape[0]), int(im.shape[1]))
    if isinstance(im, (str, bytes)) or hasattr(im, "__fspath__"):
        try:
            from PIL import Image

            with Image.open(im) as f:
                return (int(f.height), int(f.width))
        except Exception:
            return (0, 0)
    return (0, 0)


# `mlx_vlm_amd.generate` names both this module and the function (as in the reference, where the package's eager
# `from .generate import generate` rebinds the name).  The package imports lazily, so once this module is loaded the
# import system binds the MODULE on the package; make it callable so `from mlx_vlm_amd import generate` works in either order.
import sys as _sys
import types as _types


class _CallableModule(_types.ModuleType):
    def __call__(self, *args, **kwargs):
        return generate(*args, **kwargs)


_sys.modules[__name__].__class__ = _CallableModule
