"""Continuous batching over the paged KV pool: `BatchGenerator` with the reference's interface
(mlx_vlm/generate/ar.py:2178-2887: insert / next / remove / stats / has_work, `GenerationBatch.Response` 937-943,
`PromptProgress` 906-913, finish rules 1313-1316).

What differs from the reference is the mechanism, not the contract.  The reference keeps one left-padded contiguous
KV tensor per layer for the whole batch and re-packs it whenever the membership changes (`filter` / `extend`,
cache.py:1100-1201).  Here every sequence owns pages of the pool; the batch owns a small block table with one row
per batch row, so

  * a finished request leaves by releasing its pages and, if it was not the last row, the last row takes its place:
    one table row and three 4-byte state words are copied - no KV bytes move and nothing is padded;
  * a new request joins after ONE varlen prefill launch over all admitted prompts (one ViT call over all their
    images), by writing its first sampled token / position / context length into the next free row.  That admission
    (ViT + prefill + first sample) is enqueued on a SECOND HIP stream: the running rows keep decoding underneath it
    and the new rows join at the first `next()` that finds its event complete (the reference alternates decode and
    prefill on one stream, ar.py:2705-2887, so every admission stalls the running rows).  Admissions run AHEAD of
    the free rows: up to `prefill_ahead` prefilled requests wait with their pages, so a row that finishes is refilled
    in the same `next()` call instead of idling for the length of a prefill;
  * a decode step is one hipGraph replay over the first 1, 2, 4 or 8 rows (the widths the weight-streaming kernels
    are built for); rows past the live ones point at a scratch page.  The engine keeps one graph per width.

The host never waits for the step in flight: `next()` returns the tokens that were the INPUTS of the step launched
by the previous call (their copy to pinned memory was enqueued before that step), then enqueues the next step - the
reference's `mx.async_eval` double buffering (ar.py:1044-1141), with the stop decision one step behind as there.
"""
from __future__ import annotations

import contextlib
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from ._lib import h2d
from .models.cache import PAGE, PagedSequence
from .models.qwen2_vl.language import DecodeState
from .sample_utils import Sampler, make_sampler

_PROCS_KEY = "__logits_processors__"     # a request's LogitsProcessors spec inside its keyword arguments
_PYPROCS_KEY = "__py_logits_processors__"   # ... its Python callables (tokens, logits) -> logits (eager steps)
_BUDGET_KEY = "__thinking_budget_criteria__"   # ... its ThinkingBudgetCriteria
MAX_ROWS = 16         # default widest decode step: one N tile of the skinny-M MFMA GEMM (csrc/gemv_mfma.hip), bf16 or 4-bit weights
WIDE_ROWS = 64        # on request (completion_batch_size > 16): WIDE steps of 32 / 64 rows on the prefill GEMMs (engine.hip decode_impl:
                      # 192.6 us per Qwen2-VL-7B layer at 32 rows, 210 at 64, against 120 per 16-row step - profiles/r03_mfma_shapes.txt)
WIDTHS = (1, 2, 4, 8, 16, 32, 64)


class _NullEvent:
    """What the scheduler sees of a device event when no device is involved (the mock-engine tests on CPU, in the manner
    of the reference's scheduler tests with mock models, tests/test_generate.py:50-166, 529-1170)."""

    def record(self, *a):
        pass

    def query(self):
        return True

    def synchronize(self):
        pass


@dataclass
class PromptProgress:
    """reference ar.py:906-913"""
    uid: int
    prompt_tokens: int
    prompt_tps: float = 0.0
    prompt_time: float = 0.0
    cached_tokens: int = 0


@dataclass
class _Row:
    uid: int
    seq: PagedSequence
    max_tokens: int
    prompt_tokens: int
    num_tokens: int = 0
    procs: Any = None              # this request's sample_utils.LogitsProcessors (None: none)
    py_procs: Any = None           # its Python callables (tokens, logits) -> logits, in order (None: none)
    tokens: Any = None             # device int32 [n]: prompt + every token fed back (kept only for rows with py_procs)
    budget: Any = None             # its ThinkingBudgetCriteria (None: none)


@dataclass
class _Admission:
    """A prefill in flight on the side stream: everything the join needs, kept alive until then."""
    batch: list
    caches: list
    lens: List[int]
    tok0: torch.Tensor
    lp0: Optional[torch.Tensor]
    state: torch.Tensor            # int32 [2, n]: rope position, context length
    event: torch.cuda.Event
    tic: float
    removed: set = field(default_factory=set)
    joined: int = 0                # requests of `batch` already given a row (a join takes as many as there are free rows)
    done_t: Optional[float] = None # host clock when the prefill's event was first seen fired (it may WAIT for a row after that)
    pen: Any = None                # device (hist, hist_len, params, bias_idx, bias_val) of the admitted requests, or None

    def waiting(self) -> int:
        return sum(1 for b in self.batch[self.joined:] if b[0] not in self.removed)


_ADMISSION_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _admission_stream(dev) -> "torch.cuda.Stream":
    """ONE side stream per device for the life of the process: the caching allocator keeps a pool per stream, so a fresh
    stream per generator would find no cached blocks and pay ~13 hipMalloc calls (tens of ms each) in its first admissions
    (profiles/r02_continuous_diag.txt)."""
    idx = torch.device(dev).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _ADMISSION_STREAMS:
        _ADMISSION_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _ADMISSION_STREAMS[idx]


class BatchGenerator:
    """`insert(prompts)` queues token-id prompts (shortest first, ar.py:2620-2623); every `next()` reports one token
    per running request and advances the batch by one step.  `model` is the full `Model` (vision tower + language
    model): per-request `prompt_kwargs` carry `pixel_values` / `image_grid_thw` and the ViT runs inside the admission
    prefill.  Per-request logits processors (`insert(..., logits_processors=)`, ar.py:2584-2606): the reference's own
    four - logit_bias, repetition / presence / frequency penalty, as `make_logits_processors` specs - are applied by the
    device pass inside the captured step from per-row parameter tables; any other callable `(tokens, logits) -> logits`
    (and a Python `sampler(logprobs) -> tokens`) makes the steps EAGER while such a row is live: forward, the callables on
    the rows' logits as device tensors, sampling, advance - four host calls instead of one graph replay
    (ar.py:1044-1141).  `thinking_budget_criteria` per request (ar.py:1303-1350).  `kv_bits=8`: the uniform 8-bit KV
    cache with the reference's batch policy (every layer but the last of a stack deeper than 2, models/cache.py:8-21);
    as in the reference's batch path `quantized_kv_start` has no effect on the uniform scheme (ar.py:776-812: the batch
    caches are quantised from the first token).  APC and speculative drafts are outside the built path and are rejected."""

    @dataclass
    class Response:
        uid: int
        token: int
        token_logprob: float
        finish_reason: Optional[str]
        top_logprobs: Optional[List[Tuple[int, float]]] = None

    def __init__(self, model, processor=None, *, max_tokens: int = 128, stop_tokens=None,
                 sampler: Optional[Sampler] = None, completion_batch_size: int = MAX_ROWS,
                 prefill_batch_size: int = MAX_ROWS, compute_logprobs: bool = True, use_graph: bool = True,
                 async_prefill: bool = True, prefill_ahead: int = 2, **kwargs):
        # uniform 8-bit KV cache (reference ar.py:2200-2230: BatchGenerator(kv_bits=, kv_group_size=, quantized_kv_start=)):
        # every admitted request's cache is quantised right after its prefill
        self.kv_bits = kwargs.pop("kv_bits", None)
        kv_group_size = kwargs.pop("kv_group_size", None) or 64
        kv_start = kwargs.pop("quantized_kv_start", None)
        kv_scheme = kwargs.pop("kv_quant_scheme", None)
        if self.kv_bits is not None:
            if float(self.kv_bits) != 8 or int(kv_group_size) != 64 or kv_scheme not in (None, "uniform"):
                raise NotImplementedError(f"BatchGenerator: kv_bits={self.kv_bits} kv_group_size={kv_group_size} "
                                          f"kv_quant_scheme={kv_scheme}: the uniform 8-bit / group-64 quantized KV cache is built")
            # quantized_kv_start: the reference's batch path builds BatchQuantizedKVCache for the uniform scheme whatever
            # the value (ar.py:776-812 - only the TurboQuant scheme defers on it), i.e. rows are quantised from their first
            # token; accepted and, as there, without effect
            if getattr(self.lm if hasattr(self, "lm") else model.language_model, "head_dim", 128) != 128:
                raise NotImplementedError("BatchGenerator(kv_bits=8): the 8-bit KV kernels are built for 128-wide heads")
        if kwargs.get("max_kv_size"):
            # reference ar.py:831-834 (to_batch_cache): make_prompt_cache's RotatingKVCache has keep = 4
            raise ValueError("RotatingKVCache with keep tokens is not supported.")
        unsupported = {k: v for k, v in kwargs.items() if v not in (None, False, 0, [], ())
                       and k not in ("prefill_step_size", "greedy_sampling", "stream")}
        if unsupported:
            raise NotImplementedError(f"BatchGenerator: outside the built path: {sorted(unsupported)}")
        self.model = model
        self.lm = model.language_model
        self.lm._batch_users = getattr(self.lm, "_batch_users", 0) + 1      # models with per-call global state (phi3_v's RoPE regime)
        self.processor = processor
        self.tokenizer = getattr(processor, "tokenizer", processor)
        self.max_tokens = max_tokens
        self.compute_logprobs = compute_logprobs
        self.use_graph = use_graph
        self.async_prefill = async_prefill
        self.prefill_ahead = max(0, int(prefill_ahead)) if async_prefill else 0
        max_rows = MAX_ROWS
        if int(completion_batch_size) > MAX_ROWS:      # wide steps: where the language model's engine has them
            max_rows = max(MAX_ROWS, min(WIDE_ROWS, int(getattr(self.lm, "MAX_DECODE_ROWS", MAX_ROWS))))
        pool_seqs = getattr(getattr(self.lm, "pool", None), "max_seqs", 64)
        while max_rows > 8 and pool_seqs < 2 * max_rows + 2:      # rows + admissions prefilled ahead + the scratch page each hold a sequence slot
            max_rows //= 2
        self.completion_batch_size = max(1, min(int(completion_batch_size), max_rows))
        self.prefill_batch_size = max(1, int(prefill_batch_size))
        # a sample_utils.Sampler samples on the device inside the captured step; any other callable is the reference's
        # `sampler(logprobs [n, V]) -> tokens [n]` contract and makes every step eager
        self._py_sampler = None
        if sampler is not None and (not isinstance(sampler, Sampler) or sampler.extended):
            # (a Sampler with the filters the captured step does not carry - sample_utils.Sampler.extended - is called on the
            # rows' log-probs like any callable: the same HIP kernel through vlm_sample_ex)
            if not callable(sampler):
                raise TypeError("sampler must be a mlx_vlm_amd.sample_utils.Sampler or a callable logprobs -> tokens")
            if isinstance(sampler, Sampler) and sampler.xtc_probability > 0.0:
                # the reference's apply_xtc takes its minimum over the WHOLE [rows, V] array and draws once per call
                # (sample_utils.py:371-376): across requests that is not a per-request filter; vlm_sample_ex serves one row
                raise NotImplementedError("xtc in the batch generator: the reference's filter mixes the rows of a batch; one row per call is built")
            self._py_sampler, sampler = sampler, None
        self.sampler = sampler or make_sampler()
        self._sargs = self.sampler.engine_args()
        crit = getattr(self.tokenizer, "stopping_criteria", None)
        self._stop = set(getattr(crit, "eos_token_ids", ()) or ())
        self._stop.update(int(t) for t in (stop_tokens or ()))
        self.uid_count = 0
        self._unprocessed_sequences: List[tuple] = []
        self._rows: List[_Row] = []
        self._closed = False

        # buffers are sized for the widest step that can run: 3 live rows decode inside a 4-wide step
        lm, dev, cap = self.lm, self.lm.device, next(w for w in WIDTHS if w >= self.completion_batch_size)
        self._cuda = torch.device(dev).type == "cuda"
        self._st = self._new_decode_state(cap)
        pool = lm.pool
        self._table = torch.zeros(cap, pool.max_pages, dtype=torch.int32, device=dev)
        self._scratch_seq = PagedSequence(pool)          # one page nobody reads: where idle rows write their k / v
        self._scratch_seq.reserve(1)
        self._idle_row = torch.full((pool.max_pages,), self._scratch_seq.pages[0], dtype=torch.int32, device=dev)
        self._table[:] = self._idle_row
        self._lp = torch.zeros(cap, dtype=torch.float32, device=dev)      # logprob of the token sitting in st.tok
        self._pin_tok = torch.empty(2, cap, dtype=torch.int32)
        self._pin_lp = torch.empty(2, cap, dtype=torch.float32)
        if self._cuda:
            self._pin_tok, self._pin_lp = self._pin_tok.pin_memory(), self._pin_lp.pin_memory()
        self._inflight: Optional[Tuple[int, torch.cuda.Event, List[int], float]] = None
        self._pending: List[_Admission] = []         # oldest first
        self._side = _admission_stream(dev) if (async_prefill and self._cuda) else None
        self._calls = 0
        self._idle_steps = 0
        self._width = 0

        self._prompt_tokens_counter = 0
        self._prompt_time_counter = 0.0
        self._gen_tokens_counter = 0
        self._gen_time_counter = 0.0
        self._steps_counter = 0
        # decode time = wall time with a step in flight: [launch, results back] of every step PLUS the host's share of the loop up
        # to the next launch (the steps are pipelined one ahead, so that share is hidden behind the GPU but it is decode time),
        # minus what a round spent blocked on a prefill
        self._t_results: Optional[float] = None
        self._blocked = 0.0

    # ------------------------------------------------------------------ engine hooks
    # Everything that touches the device sits behind these five methods; the scheduler (queue, admission, joins,
    # compaction, finish rules, stats) never looks past them, so it is driven by a mock engine in the CPU tests.
    def _event(self):
        return torch.cuda.Event() if self._cuda else _NullEvent()

    def _new_decode_state(self, cap: int):
        return self._borrow_state(self.lm, cap)

    def _prefill_requests(self, batch):
        """One ViT call over all images of `batch`, one varlen prefill launch, first tokens sampled on the device.
        -> (caches, lengths, first tokens int32 [n], their log-probs f32 [n] or None, int32 [2, n] rope position /
        context length of each request's first decode step)"""
        from .generate import embed_requests

        lm = self.lm
        ids_l = [b[1] for b in batch]
        pix_l = [b[3].get("pixel_values") for b in batch]
        grid_l = [b[3].get("image_grid_thw") for b in batch]
        extras = [{k: v for k, v in b[3].items() if k not in ("pixel_values", "image_grid_thw", _PROCS_KEY, _PYPROCS_KEY, _BUDGET_KEY)}
                  for b in batch]
        emb, pos, lens, deltas = embed_requests(self.model, ids_l, pix_l, grid_l, extras)
        caches = [lm.make_cache() for _ in batch]
        for c, L, b in zip(caches, lens, batch):
            c[0]._seq.reserve(L + b[2] + 2)          # prompt + every token it may generate + the step in flight
        logits = lm.prefill(emb, pos, caches, lens, "last")
        # per-request logits processors on the FIRST token (ar.py:360-364: `tokens` is the prompt there): a history of each
        # admitted prompt + its parameter rows, applied by the device pass; the same rows move into the decode state at the join
        self._last_pen = None
        specs = [b[3].get(_PROCS_KEY) for b in batch]
        if any(specs):
            cap = self._st.hist.shape[1]
            host_h = np.zeros((len(batch), cap), dtype=np.int32)
            host_n = np.zeros(len(batch), dtype=np.int32)
            for r, b in enumerate(batch):
                t = np.asarray(b[1], dtype=np.int64).reshape(-1)[-cap:]
                host_h[r, :len(t)] = t
                host_n[r] = len(t)
            params, bidx, bval = DecodeState.pack_row_penalties(specs, cap)
            dev = lm.device
            pen = tuple(h2d(x, dev) for x in (host_h, host_n, params, bidx, bval))
            from . import _lib
            pa = _lib.PenaltyArgs(pen[0].data_ptr(), pen[1].data_ptr(), cap, 0.0, 0, 0.0, 0, 0.0, 0, pen[3].data_ptr(),
                                  pen[4].data_ptr(), 0, pen[2].data_ptr(), DecodeState.MAX_ROW_BIAS)
            ops.apply_logit_penalties(logits, pa)
            self._last_pen = pen
        # RNG stream of the first tokens: the sampler keys its noise on (seed, step, row, index).  Decode steps count up
        # from 1 (see _borrow_state); every admission takes its own step value from a range they never reach, so no
        # (step, row) pair is used twice - neither between an admission and a decode step nor between two admissions
        self._admissions = getattr(self, "_admissions", 0) + 1
        step0 = torch.full((1,), 0x40000000 + (self._admissions & 0x3FFFFFFF), dtype=torch.int32, device=logits.device)
        for r, b in enumerate(batch):                   # Python callables on the first token: `tokens` = the prompt (ar.py:1959-1970)
            for proc in b[3].get(_PYPROCS_KEY) or ():
                tk = h2d(np.asarray(b[1], dtype=np.int32).reshape(-1), logits.device)
                logits[r:r + 1].copy_(self._call_proc(proc, tk, logits[r:r + 1]))
        if self._py_sampler is not None:
            _, lp = ops.sample(logits, want_logprobs=True, temperature=0.0)
            tok0 = self._call_sampler(lp)
        else:
            tok0, lp = ops.sample(logits, step=step0, want_logprobs=self.compute_logprobs, **self._sargs)
        lp0 = lp.gather(1, tok0.long()[:, None]).reshape(-1).float() if self.compute_logprobs else None
        ctx = np.asarray(lens, dtype=np.int32)
        state = h2d(np.stack([ctx + np.asarray(deltas, dtype=np.int32), ctx]), lm.device)
        return caches, list(lens), tok0, lp0, state

    def _decode_rows(self, width: int):
        """One decode step over rows 0..width-1 of the state: tok <- sampled token, pos and ctx advanced by one."""
        if self._py_sampler is not None or any(row.py_procs for row in self._rows):
            return self._decode_rows_eager(width)
        self.lm.decode_step_rows(self._st, width, self._table, self._sargs, use_graph=self.use_graph,
                                 with_logprobs=self.compute_logprobs, row_penalties=any(row.procs for row in self._rows),
                                 q8=self.kv_bits is not None)

    @staticmethod
    def _call_proc(proc, tokens, logits_row):
        out = proc(tokens, logits_row)
        if not isinstance(out, torch.Tensor):
            out = torch.as_tensor(np.asarray(out), device=logits_row.device)
        return out.to(device=logits_row.device, dtype=logits_row.dtype).reshape(1, -1)

    def _call_sampler(self, logprobs):
        tok = self._py_sampler(logprobs)
        if not isinstance(tok, torch.Tensor):
            tok = torch.as_tensor(np.asarray(tok), device=logprobs.device)
        return tok.to(device=logprobs.device, dtype=torch.int32).reshape(-1).contiguous()

    def _decode_rows_eager(self, width: int):
        """The same step with the host in the middle (reference GenerationBatch._step, ar.py:1044-1141): forward of the rows
        (logits only), each row's device-side spec, then its Python callables on its logits row with its token context
        (prompt + every token fed, the one of this step included), log-probs, sampler (device or Python), advance."""
        lm, st, n = self.lm, self._st, len(self._rows)
        lm.decode_forward_rows(st, width, self._table, q8=self.kv_bits is not None)
        logits = st.logits[:width]
        if any(row.procs for row in self._rows):
            ops.apply_logit_penalties(logits, st.row_penalty_tables(), push_tok=st.tok[:width])
        for r, row in enumerate(self._rows):
            if row.py_procs:
                row.tokens = torch.cat([row.tokens, st.tok[r:r + 1]])
                for proc in row.py_procs:
                    logits[r:r + 1].copy_(self._call_proc(proc, row.tokens, logits[r:r + 1]))
        if self._py_sampler is not None:
            _, lp = ops.sample(logits, want_logprobs=True, temperature=0.0)
            tok = torch.zeros(width, dtype=torch.int32, device=logits.device)
            tok[:n].copy_(self._call_sampler(lp[:n]))
        else:
            tok, lp = ops.sample(logits, step=st.step, want_logprobs=self.compute_logprobs, **self._sargs)
        st.tok[:width].copy_(tok)
        if lp is not None:
            st.logprobs[:width].copy_(lp)
        lm.decode_advance_rows(st, width)

    def _force_next_token(self, r: int, token: int):
        self._st.tok[r:r + 1].copy_(h2d(np.asarray([token], dtype=np.int32), self._st.tok.device))

    def _quantize_joined(self, seq):
        """the joined request's cached prompt becomes a QuantizedKVCache (engine hook: a mock engine has nothing to convert)"""
        self.lm.quantize_kv([seq], bits=int(self.kv_bits), group_size=64)

    def _row_logprobs(self, n: int) -> torch.Tensor:
        """log-prob of the token each of the first n rows has just sampled (f32 [n])"""
        st = self._st
        return st.logprobs[:n].gather(1, st.tok[:n].long()[:, None]).reshape(-1).float()

    @staticmethod
    def _borrow_state(lm, cap: int) -> DecodeState:
        """Decode states are carved from the model's bump arena (never returned): generators re-use a parked one."""
        parked = lm.__dict__.setdefault("_batch_states", [])
        for st in parked:
            if st.B == cap and not getattr(st, "in_use", False):
                break
        else:
            st = DecodeState(lm, cap)
            parked.append(st)
        st.in_use = True
        for buf in (st.tok, st.pos, st.ctx, st.step):
            buf.zero_()
        st.step.fill_(1)          # sampling step counter of the decode steps (0 is never used: see _prefill_requests)
        return st

    # ------------------------------------------------------------------ queue (reference ar.py:2584-2670)
    def insert(self, prompts, max_tokens=None, prompt_kwargs: Optional[List[dict]] = None, logits_processors=None,
               thinking_budget_criteria=None) -> List[int]:
        """logits_processors: one entry per prompt (reference ar.py:2584-2606) - None, a `sample_utils.LogitsProcessors`
        (what `make_logits_processors` returns here) or a list holding one; they run on the device inside the step, each
        row with its own parameters and token history."""
        if max_tokens is None or isinstance(max_tokens, int):
            max_tokens = [max_tokens or self.max_tokens] * len(prompts)
        if prompt_kwargs is None:
            prompt_kwargs = [{}] * len(prompts)
        if logits_processors is None:
            logits_processors = [None] * len(prompts)
        if thinking_budget_criteria is None:
            thinking_budget_criteria = [None] * len(prompts)
        if len(thinking_budget_criteria) != len(prompts):
            raise ValueError("Insufficient number of thinking_budget_criteria provided")
        if len(max_tokens) != len(prompts) or len(prompt_kwargs) != len(prompts) or len(logits_processors) != len(prompts):
            raise ValueError("max_tokens / prompt_kwargs / logits_processors must have one entry per prompt")
        from .sample_utils import HIST_CAP, LogitsProcessors
        specs, pys = [], []
        for lp in logits_processors:
            items = [x for x in (lp if isinstance(lp, (list, tuple)) else [lp]) if x]
            sp = [x for x in items if isinstance(x, LogitsProcessors)]
            py = [x for x in items if not isinstance(x, LogitsProcessors)]
            if len(sp) > 1:
                raise NotImplementedError("one make_logits_processors spec per request")
            for f in py:
                if not callable(f):
                    raise TypeError("logits_processors entries must be make_logits_processors specs or callables (tokens, logits) -> logits")
            if sp:
                DecodeState.pack_row_penalties(sp, HIST_CAP)                      # validates (bias list length) before queueing
            specs.append(sp[0] if sp else None)
            pys.append(py or None)
        uids = []
        for p, m, kw, sp, py, crit in zip(prompts, max_tokens, prompt_kwargs, specs, pys, thinking_budget_criteria):
            ids = np.asarray(p, dtype=np.int64).reshape(-1)
            if ids.size == 0:
                raise ValueError("empty prompt")
            kw = dict(kw or {})
            if sp:
                kw[_PROCS_KEY] = sp               # travels with the request's keyword arguments (the queue item keeps its shape)
            if py:
                kw[_PYPROCS_KEY] = py
            if crit is not None:
                kw[_BUDGET_KEY] = crit
            self._unprocessed_sequences.append((self.uid_count, ids, int(m), kw))
            uids.append(self.uid_count)
            self.uid_count += 1
        self._unprocessed_sequences.sort(key=lambda x: len(x[1]))
        return uids

    def remove(self, uid) -> bool:
        for i, item in enumerate(self._unprocessed_sequences):
            if item[0] == uid:
                self._unprocessed_sequences.pop(i)
                return True
        for r, row in enumerate(self._rows):
            if row.uid == uid:
                self._drop_rows([r])
                return True
        for p in self._pending:                      # being prefilled / waiting for a row: dropped at the join
            if uid not in p.removed and any(b[0] == uid for b in p.batch[p.joined:]):
                p.removed.add(uid)
                return True
        return False

    @property
    def unprocessed_prompts(self):
        return self._unprocessed_sequences

    @property
    def has_pending_prompts(self) -> bool:
        return len(self._unprocessed_sequences) > 0 or bool(self._pending)

    @property
    def has_work(self) -> bool:
        return (bool(self._rows) or bool(self._unprocessed_sequences) or self._inflight is not None
                or any(p.waiting() for p in self._pending))

    def __len__(self):
        return len(self._rows)

    def stats(self):
        from .generate import BatchStats, _peak_gb

        s = BatchStats()
        s.prompt_tokens = self._prompt_tokens_counter
        s.prompt_time = self._prompt_time_counter
        s.prompt_tps = self._prompt_tokens_counter / self._prompt_time_counter if self._prompt_time_counter > 0 else 0
        s.generation_tokens = self._gen_tokens_counter
        s.generation_time = self._gen_time_counter
        s.generation_tps = self._gen_tokens_counter / self._gen_time_counter if self._gen_time_counter > 0 else 0
        s.peak_memory = _peak_gb()
        s.decode_steps = self._steps_counter       # (not a reference field: graph replays so far, for the roofline of a job)
        return s

    def close(self):
        if self._closed:
            return
        self._closed = True
        self.lm._batch_users = max(0, getattr(self.lm, "_batch_users", 1) - 1)
        if self._cuda:
            torch.cuda.synchronize()
        for p in self._pending:
            for c in p.caches[p.joined:]:
                c[0]._seq.release()
        self._pending = []
        for row in self._rows:
            row.seq.release()
        self._rows = []
        self._scratch_seq.release()
        self._inflight = None
        self._st.in_use = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ per-row logits processors (device tables of the state)
    def _set_row_penalties(self, r: int, pen, i: int, spec):
        """row r of the decode state <- request i of an admission (history of its prompt + its parameters), or cleared"""
        st = self._st
        if spec and pen is not None:
            st.row_penalty_tables()
            st.hist[r].copy_(pen[0][i])
            st.hist_len[r:r + 1].copy_(pen[1][i:i + 1])
            st.row_params[r].copy_(pen[2][i])
            st.row_bias_idx[r].copy_(pen[3][i])
            st.row_bias_val[r].copy_(pen[4][i])
        elif getattr(st, "row_params", None) is not None:
            st.row_params[r].zero_()                  # the previous occupant's processors must not apply to this request
            st.hist_len[r:r + 1].zero_()

    def _move_row_penalties(self, dst: int, src: int):
        st = self._st
        if getattr(st, "row_params", None) is not None:
            for buf in (st.hist, st.row_params, st.row_bias_idx, st.row_bias_val):
                buf[dst].copy_(buf[src])
            st.hist_len[dst:dst + 1].copy_(st.hist_len[src:src + 1])

    # ------------------------------------------------------------------ membership
    def _drop_rows(self, gone: List[int]):
        """Release the sequences in batch rows `gone`; keep the live rows dense by moving rows from the end into the
        holes (stream-ordered after the step in flight)."""
        st = self._st
        gone_set = set(gone)
        for r in gone:
            self._rows[r].seq.release()
        live_after = len(self._rows) - len(gone)
        movers = [r for r in range(live_after, len(self._rows)) if r not in gone_set]
        holes = sorted(r for r in gone if r < live_after)
        for dst, src in zip(holes, movers):
            for buf in (st.tok, st.pos, st.ctx, self._lp):
                buf[dst:dst + 1].copy_(buf[src:src + 1])
            self._table[dst].copy_(self._table[src])
            self._move_row_penalties(dst, src)
            self._rows[dst] = self._rows[src]
        del self._rows[live_after:]
        self._table[live_after:] = self._idle_row
        self._park_idle_rows()

    def _park_idle_rows(self):
        """Rows past the live ones keep running inside a 2/4/8-wide step: context 0 on the scratch page."""
        n = len(self._rows)
        if n < self._st.B:
            self._st.ctx[n:].zero_()
            self._st.pos[n:].zero_()
            self._st.tok[n:].zero_()
            if getattr(self._st, "row_params", None) is not None:
                self._st.row_params[n:].zero_()
        self._idle_steps = 0

    def _admit_begin(self):
        """Enqueue the prefill of as many queued prompts as there are free rows (one ViT call over all their images, one
        varlen prefill launch, first tokens sampled on the device) - on the side stream when async_prefill."""
        free = self.completion_batch_size - len(self._rows)
        ahead = sum(p.waiting() for p in self._pending)
        n = min(free + self.prefill_ahead - ahead, self.prefill_batch_size, len(self._unprocessed_sequences))
        if n <= 0:
            return
        batch, self._unprocessed_sequences = self._unprocessed_sequences[:n], self._unprocessed_sequences[n:]
        tic = time.perf_counter()
        where = contextlib.nullcontext()
        if self._side is not None:
            # pages released so far may still be written by steps already enqueued on the main stream
            gate = torch.cuda.Event()
            gate.record()
            self._side.wait_event(gate)
            where = torch.cuda.stream(self._side)
        with where:
            caches, lens, tok0, lp0, state = self._prefill_requests(batch)
            ev = self._event()
            ev.record()
        self._pending.append(_Admission(batch, caches, lens, tok0, lp0, state, ev, tic, pen=getattr(self, "_last_pen", None)))
        self._last_pen = None

    def _admit_join(self) -> List[PromptProgress]:
        """Give free rows to prefilled requests, oldest admission first; an admission whose event has not fired is
        waited for only when there is nothing to decode meanwhile."""
        st, lm = self._st, self.lm
        out: List[PromptProgress] = []
        while self._pending and len(self._rows) < self.completion_batch_size:
            p = self._pending[0]
            if not p.event.query():
                if self._side is not None and (self._rows or out):
                    break                             # keep decoding; it joins at a later round
                t_w = time.perf_counter()
                p.event.synchronize()                 # synchronous mode, or nothing to decode meanwhile
                self._blocked += time.perf_counter() - t_w
            if p.done_t is None:
                p.done_t = time.perf_counter()
            if self._cuda:
                torch.cuda.current_stream().wait_event(p.event)
            # wall time to the first token, as the reference reports it - up to the moment the prefill was seen complete: an
            # admission prefilled AHEAD under the decode steps then waits for rows, which is not prompt time
            dt = p.done_t - p.tic
            if p.joined == 0:
                self._prompt_tokens_counter += int(sum(p.lens))
                self._prompt_time_counter += dt
            while p.joined < len(p.batch) and len(self._rows) < self.completion_batch_size:
                i = p.joined
                p.joined += 1
                b, L, seq = p.batch[i], p.lens[i], p.caches[i][0]._seq
                if b[0] in p.removed:
                    seq.release()
                    continue
                r = len(self._rows)
                st.tok[r:r + 1].copy_(p.tok0[i:i + 1])
                st.pos[r:r + 1].copy_(p.state[0, i:i + 1])
                st.ctx[r:r + 1].copy_(p.state[1, i:i + 1])
                if p.lp0 is not None:
                    self._lp[r:r + 1].copy_(p.lp0[i:i + 1])
                self._table[r].copy_(lm.pool.block_table[seq.seq])
                if self.kv_bits is not None:
                    self._quantize_joined(seq)          # KVCache.to_quantized right after the prefill (stream-ordered)
                spec = b[3].get(_PROCS_KEY)
                self._set_row_penalties(r, p.pen, i, spec)
                py = b[3].get(_PYPROCS_KEY)
                self._rows.append(_Row(uid=b[0], seq=seq, max_tokens=b[2], prompt_tokens=L, procs=spec, py_procs=py,
                                       tokens=h2d(np.asarray(b[1], dtype=np.int32).reshape(-1), lm.device) if py else None,
                                       budget=b[3].get(_BUDGET_KEY)))
                out.append(PromptProgress(uid=b[0], prompt_tokens=L, prompt_tps=L / dt if dt > 0 else 0.0, prompt_time=dt))
            if p.joined >= len(p.batch):
                self._pending.pop(0)
            elif not p.waiting():                     # only removed requests left
                for c in p.caches[p.joined:]:
                    c[0]._seq.release()
                self._pending.pop(0)
        return out

    # ------------------------------------------------------------------ one scheduling round
    def next(self):
        """-> (prompt_responses, generation_responses), reference ar.py:2705-2887."""
        responses: List[BatchGenerator.Response] = []
        if self._inflight is not None:
            slot, ev, uids, t_launch = self._inflight
            self._inflight = None
            ev.synchronize()
            self._t_results = time.perf_counter()
            self._gen_time_counter += self._t_results - t_launch
            toks, lps = self._pin_tok[slot].numpy(), self._pin_lp[slot].numpy()
            live = {row.uid: r for r, row in enumerate(self._rows)}
            gone = []
            for i, uid in enumerate(uids):
                r = live.get(uid)
                if r is None:                      # removed by the caller since the snapshot
                    continue
                row = self._rows[r]
                row.num_tokens += 1
                tok = int(toks[i])
                reason = "stop" if tok in self._stop else "length" if row.num_tokens >= row.max_tokens else None
                if row.budget is not None:
                    # ar.py:1306-1314,1338-1347: the criteria sees the reported token; a pending forced id REPLACES the next
                    # token of the row - already sampled by the step in flight, so the write is stream-ordered behind it
                    row.budget(tok)
                    forced = row.budget.pop_forced_token_id()
                    if forced is not None and reason is None:
                        self._force_next_token(r, int(forced))
                if reason is not None:
                    gone.append(r)
                responses.append(self.Response(uid, tok, float(lps[i]) if self.compute_logprobs else 0.0, reason))
            self._gen_tokens_counter += len(responses)
            if gone:
                self._drop_rows(gone)
        now = time.perf_counter()
        for p in self._pending:                      # (polled once per round: the granularity of prompt_time is one decode step)
            if p.done_t is None and p.event.query():
                p.done_t = now
        if self._side is None:
            self._admit_begin()                      # synchronous mode: prefill, join, then decode (the reference's order)
            self._blocked += time.perf_counter() - now
        prompt_responses = self._admit_join()
        if self._rows:
            self._launch_step()
        self._t_results, self._blocked = None, 0.0
        if self._side is not None:
            self._admit_begin()                      # under the step just enqueued
        return prompt_responses, responses

    def _launch_step(self):
        st, n = self._st, len(self._rows)
        width = next(w for w in WIDTHS if w >= n)
        if width != self._width or self._idle_steps >= 4 * PAGE:
            self._width = width
            self._park_idle_rows()
        slot = self._calls & 1
        self._calls += 1
        self._pin_tok[slot, :n].copy_(st.tok[:n], non_blocking=True)
        if self.compute_logprobs:
            self._pin_lp[slot, :n].copy_(self._lp[:n], non_blocking=True)
        ev = self._event()
        ev.record()
        t_launch = time.perf_counter()
        if self._t_results is not None:              # the loop went straight from the last step's results to this launch
            self._gen_time_counter += max(0.0, t_launch - self._t_results - self._blocked)
        self._inflight = (slot, ev, [row.uid for row in self._rows], t_launch)
        longest = max(row.prompt_tokens + row.max_tokens for row in self._rows) + 2
        # (16-row steps keep the single-pass attention: the split merge lives in the o_proj prologue of the <= 8-row GEMV)
        st.nsplit = 1 if longest <= 2048 or width > 8 else max(2, min(32, (longest + 16 * PAGE - 1) // (16 * PAGE)))
        self._decode_rows(width)
        if self.compute_logprobs:
            self._lp[:n].copy_(self._row_logprobs(n))
        for row in self._rows:
            row.seq.offset += 1
        self._idle_steps += 1
        self._steps_counter += 1


def generate_batch_continuous(model, input_ids_list, pixel_values_list, grids, *, max_tokens=128, stop_ids=(),
                              sampler: Optional[Sampler] = None, batch_size: int = MAX_ROWS, use_graph: bool = True,
                              extras: Optional[List[Optional[dict]]] = None, logits_processors=None, kv_bits=None):
    """The reference's `_generate_batch` loop (ar.py:3212-3232) over the continuous generator: every request is queued
    at once, the generator keeps up to `batch_size` of them decoding and admits the next ones as rows free up.
    extras: per-request keyword arguments of the model's `get_input_embeddings` besides pixels / grid (phi3_v: image_sizes).
    logits_processors: one `make_logits_processors` spec (or None) per request, or ONE spec for all of them.
    kv_bits: 8 = every request's cache is a QuantizedKVCache from its first decode step on (BatchGenerator(kv_bits=8)).
    -> (tokens per request without the stop token, BatchStats)"""
    gen = BatchGenerator(model, None, max_tokens=max(max_tokens) if isinstance(max_tokens, (list, tuple)) else max_tokens,
                         stop_tokens=set(stop_ids), sampler=sampler,
                         completion_batch_size=batch_size, prefill_batch_size=batch_size, compute_logprobs=False,
                         use_graph=use_graph, kv_bits=kv_bits)
    kw: List[Dict[str, Any]] = [dict(pixel_values=p, image_grid_thw=g) if p is not None else {}
                                for p, g in zip(pixel_values_list, grids)]
    for k, e in zip(kw, extras or []):
        if e and "pixel_values" in k:
            k.update(e)
            if k.get("image_grid_thw") is None:
                del k["image_grid_thw"]
    if logits_processors is not None and not isinstance(logits_processors, (list, tuple)):
        logits_processors = [logits_processors] * len(input_ids_list)
    uids = gen.insert([np.asarray(i).reshape(-1) for i in input_ids_list], max_tokens, prompt_kwargs=kw,
                      logits_processors=logits_processors)
    results = {u: [] for u in uids}
    tic = time.perf_counter()
    while gen.has_work:
        _, out = gen.next()
        for r in out:
            if r.finish_reason != "stop":
                results[r.uid].append(r.token)
    total = time.perf_counter() - tic
    stats = gen.stats()               # generation_time: wall time with decode steps in flight (prefills admitted under them included)
    gen.close()
    stats.wall_time = total
    return [results[u] for u in uids], stats
