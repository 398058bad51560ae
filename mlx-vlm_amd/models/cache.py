"""Paged KV cache for the MI355X engine, with the reference's cache facade.

The reference keeps one contiguous [B, Hkv, S, D] K and V per layer and grows
it by 256-token steps with zeros+concat (mlx_vlm/models/cache.py:337-439
KVCache; make_prompt_cache cache.py:45-70).  Here ALL layers of ALL sequences
share two preallocated pools (288 GB of HBM make preallocation the natural
choice) addressed through a block table, 64 tokens per page:

    K pool [layer][page][Hkv][D/8][64][8]      V pool [layer][page][Hkv][D][64 key slots]

The objects handed to user code keep the reference's contract (`offset`,
`state`, `is_trimmable/trim`, `size`, `empty`, `nbytes`, one object per layer
from `make_prompt_cache`), so a `prompt_cache=` can be passed in and out of
generate_step as in the reference (GenerateKwargs.prompt_cache).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from .._lib import h2d

PAGE = 64
# V key-slot order inside a page (csrc/common.hpp vlm_vslot): slot of token `w` of the page
VSLOT = [(w & 32) + 8 * (((w & 31) & 15) >> 2) + 4 * ((w & 31) >> 4) + (w & 3) for w in range(PAGE)]


class Arena:
    """One contiguous device allocation carved into the engine's SMALL buffers (decode state, block table, norm
    weights, biases, ...).  Measured on MI355X: the first access of a kernel to each separately allocated small
    buffer costs ~4.5 us (address-translation miss), and a decode kernel touches several of them in a dependent
    chain; with everything small inside one 2 MB-aligned region a kernel pays that once."""

    def __init__(self, nbytes: int, device="cuda", zero: bool = True):
        self.buf = (torch.zeros if zero else torch.empty)(nbytes, dtype=torch.uint8, device=device)
        self.off = 0

    def alloc(self, shape, dtype, zero: bool = False, align: int = 256) -> torch.Tensor:
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        start = (self.off + align - 1) // align * align
        if start + n > self.buf.numel():
            raise RuntimeError(f"Arena exhausted: need {n} bytes at {start} of {self.buf.numel()}")
        self.off = start + n
        t = self.buf[start:start + n].view(dtype).view(shape)
        if zero:
            t.zero_()
        return t

    def put(self, src: torch.Tensor) -> torch.Tensor:
        t = self.alloc(src.shape, src.dtype)
        t.copy_(src)
        return t


class KVPool:
    """Device pools + page allocator + block table for one language model."""

    IDENTITY_BUDGET = 64 << 30   # bytes of KV the "auto" layout may reserve for the identity mapping (288 GB of HBM per GPU:
                                 # the benchmark's 40 sequence slots x 32 Ki tokens of a 2B model are 37 GB)

    def __init__(self, n_layers: int, n_kv_heads: int, head_dim: int, max_tokens: int = 32768, max_seqs: int = 64,
                 max_pages_per_seq: Optional[int] = None, device="cuda", dtype=torch.bfloat16,
                 arena: Optional["Arena"] = None, layout: str = "auto"):
        """layout: "paged"    - pages come from a shared free list, kernels walk the block table;
                   "identity" - sequence slot s owns pages [s * max_pages, (s + 1) * max_pages): the block table is
                                still filled (prefill and generic callers use it) but the decode kernels are told
                                so (block_table = NULL) and compute page numbers instead of loading them - one
                                dependent memory round trip less per decode attention / KV write;
                   "auto"     - identity when max_seqs * max_pages of KV fits IDENTITY_BUDGET."""
        self.n_layers, self.n_kv_heads, self.head_dim = n_layers, n_kv_heads, head_dim
        self.n_pages = (max_tokens + PAGE - 1) // PAGE
        self.max_seqs = max_seqs
        self.max_pages = max_pages_per_seq or self.n_pages
        page_bytes = 2 * n_layers * n_kv_heads * PAGE * head_dim * torch.empty((), dtype=dtype).element_size()
        if layout == "auto":
            # ... and never more than a third of what the DEVICE has free right now (weights are already resident when the
            # pool is built; ensure_q8 may later add half of it again; several ranks per GPU or a small part must fall
            # back to the shared free list instead of failing at load)
            budget = self.IDENTITY_BUDGET
            try:
                if torch.cuda.is_available() and torch.device(device).type == "cuda":
                    budget = min(budget, torch.cuda.mem_get_info(torch.device(device))[0] // 3)
            except Exception:       # no device (host-side construction in the CPU tests)
                pass
            layout = "identity" if max_seqs * self.max_pages * page_bytes <= budget else "paged"
        if layout not in ("identity", "paged"):
            raise ValueError(f"KVPool layout {layout!r}")
        self.identity = layout == "identity"
        if self.identity:
            self.n_pages = max_seqs * self.max_pages
        self.device = device
        per_layer = self.n_pages * n_kv_heads * PAGE * head_dim
        # zero-filled: never-written slots must not hold NaN bit patterns
        self.kpool = torch.zeros(n_layers, per_layer, dtype=dtype, device=device)
        self.vpool = torch.zeros(n_layers, per_layer, dtype=dtype, device=device)
        self.layer_stride = per_layer
        self.block_table_host = np.zeros((max_seqs, self.max_pages), dtype=np.int32)
        self.block_table = (arena.alloc((max_seqs, self.max_pages), torch.int32, zero=True) if arena is not None
                            else torch.zeros(max_seqs, self.max_pages, dtype=torch.int32, device=device))
        self._free_pages = list(range(self.n_pages - 1, -1, -1))
        self._free_seqs = set(range(max_seqs))

    # ---- allocation (host side; the block table row is re-uploaded when it changes)
    def new_seq(self) -> int:
        return self.new_seqs(1)[0]

    def new_seqs(self, n: int) -> List[int]:
        """n CONSECUTIVE block-table rows (a decode batch indexes rows by batch position)."""
        free = sorted(self._free_seqs)
        for i in range(len(free) - n + 1):
            if free[i + n - 1] - free[i] == n - 1:
                rows = free[i:i + n]
                self._free_seqs.difference_update(rows)
                return rows
        raise RuntimeError(f"KVPool: no {n} consecutive free sequence slots")

    def free_seq(self, seq: int, pages: List[int]):
        if not self.identity:
            self._free_pages.extend(pages)
        self._free_seqs.add(seq)

    def ensure(self, seq: int, pages: List[int], n_tokens: int):
        """Make sure `seq` owns pages for n_tokens tokens."""
        need = (n_tokens + PAGE - 1) // PAGE
        if need > self.max_pages:
            raise RuntimeError(f"KVPool: sequence needs {need} pages > max_pages_per_seq {self.max_pages}")
        grew = False
        while len(pages) < need:
            if self.identity:
                p = seq * self.max_pages + len(pages)
            elif not self._free_pages:
                raise RuntimeError("KVPool: out of KV pages")
            else:
                p = self._free_pages.pop()
            self.block_table_host[seq, len(pages)] = p
            pages.append(p)
            grew = True
        if grew:
            self.block_table[seq].copy_(h2d(self.block_table_host[seq].copy(), self.block_table.device))

    # ---- uniform 8-bit KV cache (reference QuantizedKVCache, cache.py:233-334): pools allocated on first use
    kpool8 = vpool8 = ksb = vsb = None

    def ensure_q8(self):
        """The 8-bit pools next to the bf16 ones: same pages, same block table (include/vlm_hip.h, vlm_attn_decode_paged_q8):
        u8 K / V with the bf16 layouts at 1 byte per element + one (scale | bias) word per (key, 64-wide group)."""
        if self.kpool8 is None:
            if self.head_dim != 128:
                raise NotImplementedError("8-bit KV cache: head_dim 128 (the engine's decode layout)")
            n = self.layer_stride
            self.kpool8 = torch.zeros(self.n_layers, n, dtype=torch.uint8, device=self.device)
            self.vpool8 = torch.zeros(self.n_layers, n, dtype=torch.uint8, device=self.device)
            self.ksb = torch.zeros(self.n_layers, n // 64, dtype=torch.int32, device=self.device)
            self.vsb = torch.zeros(self.n_layers, n // 64, dtype=torch.int32, device=self.device)
        return self

    @property
    def nbytes(self):
        return self.kpool.numel() * self.kpool.element_size() * 2

    def layer_views(self, layer: int):
        H, D = self.n_kv_heads, self.head_dim
        k = self.kpool[layer].view(self.n_pages, H, D // 8, PAGE, 8)
        v = self.vpool[layer].view(self.n_pages, H, D, PAGE)
        return k, v


class PagedSequence:
    """One sequence's share of the pool (all layers advance together)."""

    def __init__(self, pool: KVPool, seq: Optional[int] = None):
        self.pool = pool
        self.seq = pool.new_seq() if seq is None else seq
        self.pages: List[int] = []
        self.offset = 0          # tokens stored (== the reference's KVCache.offset)
        # KVCache.update_and_fetch appends to ONE layer (the reference's caches are per-layer objects, cache.py:345-367): while a
        # forward walks the layers their counts differ; `offset` (what the engine's own paths use) moves when every layer has the
        # token (advance_layer).  None = all layers level with `offset`.
        self._layer_off: Optional[List[int]] = None
        self.released = False
        self.q8 = False          # True once the sequence's cache has become a QuantizedKVCache (LanguageModel.quantize_kv)
        # max_kv_size (reference RotatingKVCache, cache.py:442-625): None = unbounded.  `held` = entries the pool holds for this
        # sequence (== offset until the window is full), `ring` = the non-sink slots in the age order of their tokens
        self.max_size: Optional[int] = None
        self.keep = 0
        self.held = 0
        self.ring = None
        self.ring_idx = 0        # the reference's `_idx` (what its Qwen2-VL reads as the cache offset, language.py:426-431)

    # ------------------------------------------------------------------ per-layer appends (update_and_fetch)
    def layer_offset(self, layer: int) -> int:
        return self.offset if self._layer_off is None else self._layer_off[layer]

    def advance_layer(self, layer: int, n: int):
        if self._layer_off is None:
            self._layer_off = [self.offset] * self.pool.n_layers
        self._layer_off[layer] += n
        lo = min(self._layer_off)
        self.offset = lo
        if lo == max(self._layer_off):
            self._layer_off = None

    def set_offset(self, v: int):
        """every layer at `v` tokens (trim, a caller's `cache.offset = n`)"""
        self.offset = int(v)
        self._layer_off = None

    # ------------------------------------------------------------------ max_kv_size
    def set_rotating(self, max_size: int, keep: int = 4):
        if self.offset:
            raise ValueError("max_kv_size is set on an empty cache")
        if int(max_size) <= keep + 1:
            raise ValueError(f"max_kv_size has to exceed keep + 1 = {keep + 1} (the window holds the sinks, the staging slot and "
                             f"at least one more token), got {max_size}")
        self.max_size, self.keep = int(max_size), int(keep)

    @property
    def rotating(self) -> bool:
        return self.max_size is not None

    @property
    def kv_entries(self) -> int:
        """entries a decode step finds in the pool (the engine writes its token at this slot and attends over one more)"""
        return self.held if self.rotating else self.offset

    @property
    def rope_offset(self) -> int:
        """the offset a family that reads `cache._idx` adds its rope delta to (Qwen2-VL); the others use `offset`"""
        return self.ring_idx if self.rotating else self.offset

    def note_prefill(self, n_tokens: int):
        """after a prompt of n_tokens landed at slots [held, held + n): the reference's _update_concat keeps a first prompt whole
        (cache.py:486-505); a SECOND multi-token update would trim the buffer to max_size - 1 + S first - not built"""
        if not self.rotating:
            return
        if self.held:
            raise NotImplementedError("max_kv_size: a second multi-token update of the rotating cache (chunked prefill / a "
                                      "prompt-cache continuation) trims the window in the reference; only the first prompt is built")
        self.held = self.ring_idx = n_tokens

    def rotate_plan(self):
        """What has to move before the next ONE-token step so that the pool holds what the reference's _update_in_place leaves
        (cache.py:507-547) minus the token about to be written: -> (src_slots, dst_slots) or None.  Updates held / ring /
        ring_idx as the reference's trim + wrap do."""
        if not self.rotating:
            return None
        M, K = self.max_size, self.keep
        plan = None
        if self.held > M:
            # a prompt longer than the window: the reference cuts the buffer to keep + the most recent M - keep and then
            # overwrites the oldest of those: sinks + the last M - keep - 1 tokens stay; survivors beyond slot M - 2 move into
            # the holes below it (disjoint sets, any pairing)
            L = self.held
            first = L - (M - K - 1)                                   # oldest surviving non-sink token (slot == token index)
            stay = [t for t in range(first, L) if t < M - 1]
            src = [t for t in range(first, L) if t >= M - 1]
            holes = [sl for sl in range(K, M - 1) if sl < first]
            assert len(src) == len(holes), (len(src), len(holes))
            where = {t: t for t in stay}
            where.update(dict(zip(src, holes)))
            self.ring = [where[t] for t in range(first, L)]           # age order
            self.held = M - 1
            self.ring_idx = K                                         # trim -> _idx = max_size -> wraps to keep
            plan = (src, holes)
        elif self.held == M:
            if self.ring is None:                                     # first time full: slot == token index
                self.ring = list(range(K, M - 1))
            oldest = self.ring.pop(0)
            self.ring.append(oldest)                                  # the newest token (staging slot M - 1) moves there
            self.held = M - 1
            if self.ring_idx >= M:
                self.ring_idx = K
            plan = ([M - 1], [oldest])
        return plan

    def note_decode_step(self):
        """one token was written at slot `held` (ring position ring_idx) and the offset grew"""
        if self.rotating:
            self.held += 1
            self.ring_idx += 1

    def reserve(self, n_total_tokens: int):
        if self.rotating and self.held:          # (decode: the window never grows past max_size; the prompt itself is kept whole -
            # the prefill call reserves max(prompt, max_size) + 1 itself, LanguageModel.prefill)
            n_total_tokens = min(n_total_tokens, max(self.held, self.max_size) + 1)
        self.pool.ensure(self.seq, self.pages, n_total_tokens)

    def release(self):
        if not self.released:
            self.pool.free_seq(self.seq, self.pages)
            self.pages = []
            self.released = True

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class KVCache:
    """Per-layer facade with the reference's KVCache interface (cache.py:337-439)."""

    step = 256  # kept for interface parity; pages are 64 tokens

    def __init__(self, seq: PagedSequence, layer: int):
        self._seq = seq
        self._layer = layer

    @property
    def offset(self):
        return self._seq.layer_offset(self._layer)

    @offset.setter
    def offset(self, v):
        self._seq.set_offset(int(v))

    def size(self):
        s = self._seq
        return min(s.offset, s.max_size) if s.rotating else self.offset      # RotatingKVCache.size (cache.py:554-555)

    def empty(self):
        return self.offset == 0

    def update_and_fetch(self, keys: torch.Tensor, values: torch.Tensor):
        """reference cache.py:345-367: append `keys` / `values` [1, Hkv, S, D] (device tensors) to THIS layer's cache and return
        what attention sees, (keys[..., :offset, :], values[..., :offset, :]).  The engine's own paths (prefill / decode
        launches) write K / V themselves; this is the module-contract entry for code that computes k and v on its own - a
        reference-side model file, a test.  The rows go into the sequence's pages (vlm_kv_append_tokens); the return value is
        materialised from the pages as `state` is."""
        self._append(keys, values)
        return self.state

    def _append(self, keys: torch.Tensor, values: torch.Tensor):
        """the write half of update_and_fetch (no materialisation of the row's K / V: BatchKVCache appends row by row and
        builds the padded batch state once)"""
        from .. import _lib
        from .._lib import check
        import ctypes as C

        s = self._seq
        if s.rotating or s.q8:
            raise NotImplementedError("update_and_fetch on a rotating / quantized cache: the engine's decode step maintains those")
        pool = s.pool
        if keys.dim() != 4 or values.shape != keys.shape or keys.shape[0] != 1 or keys.shape[1] != pool.n_kv_heads \
                or keys.shape[3] != pool.head_dim:
            raise ValueError(f"update_and_fetch: keys / values [1, {pool.n_kv_heads}, S, {pool.head_dim}], got {tuple(keys.shape)} / "
                             f"{tuple(values.shape)}")
        if keys.device.type != "cuda":
            raise RuntimeError("update_and_fetch: device tensors (the K / V pools live in HBM; there is no host path)")
        S = int(keys.shape[2])
        prev = self.offset
        if S:
            k = keys[0].to(pool.kpool.dtype)
            v = values[0].to(pool.vpool.dtype)
            if k.stride(2) != 1 or k.stride(0) % 8 or k.stride(1) % 8 or k.data_ptr() % 16:
                k = k.contiguous()
            if v.stride(2) != 1:
                v = v.contiguous()
            s.reserve(prev + S)
            esz = pool.kpool.element_size()
            lo = self._layer * pool.layer_stride * esz
            bt = None if pool.identity else pool.block_table.data_ptr()
            check(_lib.lib().vlm_kv_append_tokens(pool.kpool.data_ptr() + lo, pool.vpool.data_ptr() + lo, k.data_ptr(), v.data_ptr(), S,
                                                  k.stride(0), k.stride(1), v.stride(0), v.stride(1), s.seq, prev, bt, pool.max_pages,
                                                  pool.n_kv_heads, pool.head_dim,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "kv_append_tokens")
            self._keep_uaf = (k, v)             # alive until the stream has run
            s.advance_layer(self._layer, S)

    def extract(self, idx: int):
        """reference cache.py:395-413: row `idx` of a (one-row) cache as a KVCache of its own.  The reference copies the row; here
        the result shares the sequence's pages (a facade over the same PagedSequence)."""
        if idx not in (0, -1):
            raise IndexError(f"KVCache row index {idx} out of range for batch size 1")
        return KVCache(self._seq, self._layer)

    @classmethod
    def merge(cls, caches):
        return BatchKVCache.merge(caches)

    @property
    def max_size(self):
        return self._seq.max_size

    @property
    def keep(self):
        return self._seq.keep

    def is_trimmable(self):
        s = self._seq
        return s.offset < s.max_size if s.rotating else True              # cache.py:574-575

    def trim(self, n):
        # all layer views share the sequence: only layer 0 moves the offset so that
        # `for c in cache: c.trim(n)` (reference dispatch.py:868-870) trims once
        s = self._seq
        n = min(s.offset, n)
        if self._layer == 0:
            if s.rotating and n and s.ring is not None:
                raise NotImplementedError("trim of a rotating cache whose window has wrapped (the reference moves offset and _idx "
                                          "only, cache.py:577-581: its buffer then holds tokens the offset no longer counts)")
            s.set_offset(s.offset - n)
            if s.rotating:
                s.held -= n
                s.ring_idx -= n
        return n

    @property
    def state(self):
        """Materialise contiguous (keys, values) [1, Hkv, S, D] from the pages (debug / interop).  A rotating window: the
        entries held, in SLOT order (the reference's buffer is in ring order; attention sees the same set either way)."""
        pool = self._seq.pool
        S = self._seq.kv_entries if self._seq.rotating else self.offset
        kp, vp = pool.layer_views(self._layer)
        H, D = pool.n_kv_heads, pool.head_dim
        if S == 0:
            z = torch.zeros(1, H, 0, D, dtype=kp.dtype, device=kp.device)
            return z, z.clone()
        pages = torch.tensor(self._seq.pages, dtype=torch.long, device=kp.device)
        k = kp[pages]                       # [np, H, D/8, 64, 8]
        k = k.permute(1, 0, 3, 2, 4).reshape(H, -1, D)[:, :S]
        slots = torch.tensor(VSLOT, dtype=torch.long, device=vp.device)
        v = vp[pages][..., slots].permute(1, 0, 3, 2).reshape(H, -1, D)[:, :S]      # [np,H,D,64] -> token order
        return k[None].contiguous(), v[None].contiguous()

    @property
    def nbytes(self):
        return len(self._seq.pages) * PAGE * self._seq.pool.n_kv_heads * self._seq.pool.head_dim * 2 * 2

    def make_mask(self, N, return_array=False, window_size=None):
        return None if N == 1 else "causal"


class BatchKVCache:
    """The reference's BatchKVCache (cache.py:972-1201) for ONE layer, over rows of the paged pool.

    The reference keeps a left-padded [B, Hkv, S, D] tensor per layer and its batch operations copy it: `filter` gathers rows,
    `extend` pads and concatenates, `merge` builds a padded tensor from single caches, `finalize` rolls right padding to the
    left.  Here a row IS a sequence of the pool (its pages are named by a block-table row) and no padding is ever stored, so
    the same operations are bookkeeping: `filter` / `extend` / `merge` re-list the rows, `finalize` drops the padded tail
    of a row, `prepare(left_padding=)` only records the numbers the reference's masks would use.  The numbers the reference
    exposes keep their meaning - `_idx` (length of the padded window), `left_padding[i]`, `offset[i] = _idx - left_padding[i]`
    (tokens row i really holds) - and are pinned to the reference's own class run over the shim
    (tests/golden/make_golden_batchcache.py).  All layers of a row share one PagedSequence; like `KVCache.trim`, the
    operations that change how many tokens a row holds act on the sequence once (layer 0's object).
    """

    step = 256

    def __init__(self, left_padding: List[int], pool: Optional[KVPool] = None, layer: int = 0, rows: Optional[List[PagedSequence]] = None):
        """left_padding as the reference (cache.py:975-1000); `pool` (or `rows`) names the paged pool the rows live in - the
        reference's constructor needs no such thing because its tensors appear at the first update."""
        self.left_padding = np.asarray(list(left_padding), dtype=np.int64)
        self.offset = -self.left_padding.copy()
        self._idx = 0
        self._right_padding = None
        self._layer = int(layer)
        self._pool = pool if pool is not None else (rows[0].pool if rows else None)
        self._rows: List[Optional[PagedSequence]] = list(rows) if rows is not None else [None] * len(self.left_padding)
        if len(self._rows) != len(self.left_padding):
            raise ValueError("BatchKVCache: one row per left_padding entry")

    @classmethod
    def for_layers(cls, pool: KVPool, left_padding: List[int]) -> List["BatchKVCache"]:
        """one BatchKVCache per layer over the SAME rows (what a model's per-layer cache list is for a batch)"""
        rows = [PagedSequence(pool) for _ in left_padding]
        return [cls(left_padding, rows=rows, layer=l) for l in range(pool.n_layers)]

    # ------------------------------------------------------------------ rows
    def _row(self, i: int) -> PagedSequence:
        if self._rows[i] is None:
            if self._pool is None:
                raise RuntimeError("BatchKVCache: no pool (construct it with pool= / rows=, or through merge / the model's make_cache)")
            self._rows[i] = PagedSequence(self._pool)
        return self._rows[i]

    def _kept(self, i: int) -> int:
        """tokens row i holds in THIS layer"""
        r = self._rows[i]
        return 0 if r is None else r.layer_offset(self._layer)

    # ------------------------------------------------------------------ reference interface
    def update_and_fetch(self, keys: torch.Tensor, values: torch.Tensor):
        """cache.py:1002-1025: keys / values [B, Hkv, S, D] land at window positions [_idx, _idx + S).  Positions left of a
        row's left padding hold padding in the reference (masked out of every attention): they are not stored.  -> the
        padded (keys, values) [B, Hkv, _idx, D] as the reference returns them (zeros where it holds padding)."""
        B, S = int(keys.shape[0]), int(keys.shape[2])
        if B != len(self._rows):
            raise ValueError(f"update_and_fetch: {B} rows for a cache of {len(self._rows)}")
        for i in range(B):
            first = max(0, int(self.left_padding[i]) - self._idx)          # leading positions of this call that are padding
            if first < S:
                KVCache(self._row(i), self._layer)._append(keys[i:i + 1, :, first:], values[i:i + 1, :, first:])
        self._advance(S)
        return self.state[:2]

    def _advance(self, S: int):
        self.offset = self.offset + S
        self._idx += S

    def prepare(self, *, left_padding=None, lengths=None, right_padding=None):
        """cache.py:1027-1040"""
        if left_padding is not None:
            if self._idx != 0 or any(self._kept(i) for i in range(len(self._rows))):
                raise ValueError("Left padding can only be added to an empty BatchKVCache")
            lp = np.asarray(list(left_padding), dtype=np.int64)
            self.left_padding = self.left_padding + lp
            self.offset = self.offset - lp
        if right_padding is not None and max(right_padding) > 0:
            self._right_padding = np.asarray(list(right_padding), dtype=np.int64)

    def finalize(self):
        """cache.py:1042-1049: the reference rolls every row right by its right padding (the padded tail wraps to the front and
        becomes left padding).  Unpadded rows: the tail tokens of the right-padded call are dropped from the row."""
        if self._right_padding is not None:
            pad = self._right_padding
            if self._layer == 0:
                for i, r in enumerate(self._rows):
                    if r is not None and pad[i]:
                        r.set_offset(max(0, r.offset - int(pad[i])))
            self.offset = self.offset - pad
            self.left_padding = self.left_padding + pad
            self._right_padding = None

    @property
    def state(self):
        """(keys, values, offset, left_padding): keys / values padded [B, Hkv, _idx, D], zeros in the padding"""
        B = len(self._rows)
        pool = self._pool
        if pool is None or self._idx == 0:
            return None, None, self.offset, self.left_padding
        H, D = pool.n_kv_heads, pool.head_dim
        k = torch.zeros(B, H, self._idx, D, dtype=pool.kpool.dtype, device=pool.kpool.device)
        v = torch.zeros_like(k)
        for i, r in enumerate(self._rows):
            n = min(self._kept(i), self._idx - int(self.left_padding[i]))
            if r is not None and n > 0:
                kk, vv = KVCache(r, self._layer).state
                k[i, :, int(self.left_padding[i]): int(self.left_padding[i]) + n] = kk[0, :, :n]
                v[i, :, int(self.left_padding[i]): int(self.left_padding[i]) + n] = vv[0, :, :n]
        return k, v, self.offset, self.left_padding

    def is_trimmable(self):
        return True

    def trim(self, n):
        """cache.py:1065-1069"""
        n = min(self._idx, n)
        if self._layer == 0:
            for r in self._rows:
                if r is not None:
                    r.set_offset(max(0, r.offset - n))
        self._idx -= n
        self.offset = self.offset - n
        return n

    def make_mask(self, N: int, return_array: bool = False, **kwargs):
        # (no padding is stored: the engine's kernels take the rows' own lengths; cache.py:1071-1074 builds an array mask)
        return None if N == 1 else "causal"

    def filter(self, batch_indices):
        """cache.py:1076-1098: keep the given rows (in that order), then shift the window left by the smallest left padding"""
        idx = [int(i) for i in np.asarray(batch_indices).reshape(-1)]
        self._rows = [self._rows[i] for i in idx]
        self.offset = self.offset[idx]
        self.left_padding = self.left_padding[idx]
        if self._right_padding is not None:
            self._right_padding = self._right_padding[idx]
        min_left_pad = int(self.left_padding.min()) if len(idx) else 0
        if min_left_pad > 0:
            self._idx -= min_left_pad
            self.left_padding = self.left_padding - min_left_pad

    def extend(self, other: "BatchKVCache"):
        """cache.py:1100-1146: the rows of `other` join; both windows are right-justified at max(_idx)"""
        if self._pool is None:
            self._pool = other._pool
        max_idx = max(self._idx, other._idx)
        self.left_padding = np.concatenate([self.left_padding + (max_idx - self._idx), other.left_padding + (max_idx - other._idx)])
        self.offset = np.concatenate([self.offset, other.offset])
        self._rows = self._rows + other._rows
        self._idx = max_idx

    def extract(self, idx: int) -> "KVCache":
        """cache.py:1148-1154: row idx as a KVCache (sharing the row's pages; the reference copies)"""
        return KVCache(self._row(int(idx)), self._layer)

    @classmethod
    def merge(cls, caches: List["KVCache"]):
        """cache.py:1156-1188: single caches become the rows of a batch, right-justified at the longest.  The reference copies
        them into one padded tensor and its callers drop the singles (ar.py:743-746); here the singles' sequences BECOME the
        rows - no K / V bytes move, and a single cache kept by the caller aliases its row from then on."""
        lengths = [c.size() for c in caches]
        max_length = max(lengths)
        padding = [max_length - n for n in lengths]
        layer = caches[0]._layer
        out = cls(padding, rows=[c._seq for c in caches], layer=layer)
        if max_length:
            out.offset = out.offset + max_length
            out._idx = max_length
        return out

    def size(self):
        return self._idx

    def empty(self):
        return self._idx == 0 and not any(self._kept(i) for i in range(len(self._rows)))

    @property
    def batch_size(self):
        return len(self._rows)

    def is_single_row(self):
        return self.batch_size == 1

    @property
    def nbytes(self):
        return sum(len(r.pages) * PAGE * r.pool.n_kv_heads * r.pool.head_dim * 2 * 2 for r in self._rows if r is not None)


def make_prompt_cache(model, max_kv_size: Optional[int] = None):
    """reference cache.py:45-70.  There the built families define no `make_cache`, so `max_kv_size` gives every layer a
    RotatingKVCache(max_size=max_kv_size, keep=4); here `make_cache` is the paged pool's allocator and the bound becomes a
    property of the sequence all layer facades share (PagedSequence.set_rotating)."""
    if not hasattr(model, "make_cache"):
        raise ValueError("make_prompt_cache: the language model must provide make_cache() (paged pool owner)")
    caches = model.make_cache()
    if max_kv_size is not None:
        caches[0]._seq.set_rotating(int(max_kv_size), keep=4)
    return caches
