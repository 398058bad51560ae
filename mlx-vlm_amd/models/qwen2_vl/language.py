"""Qwen2-VL language model on MI355X - host mirror of the reference's
mlx_vlm/models/qwen2_vl/language.py (LanguageModel: same call contract,
get_rope_index, rope-delta position rule) on top of the native engine
(csrc/engine.hip): prefill = MFMA GEMMs + causal flash attention + paged KV
write, decode = weight-streaming GEMV chain + paged split-K attention, both
enqueued by ONE native call per forward.

Differences from the reference that the contract allows:
  * `supports_logits_to_keep = True` (reference hook, generate/ar.py:340-341):
    with logits_to_keep=1 only the last row goes through the lm_head instead of
    all L rows (the reference computes all and slices [:, -1], ar.py:358).
  * `make_cache()` (reference hook, cache.py:57-58) returns per-layer views of
    one paged sequence instead of 28 independent contiguous KVCache objects.
  * position ids are built on the host from host-resident ids / grids and
    uploaded once; the reference does the same work with per-token Python
    loops and .item() syncs (language.py:242-382).
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from ... import _lib, ops
from ..._lib import check
from ..base import LanguageModelOutput
from ..cache import PAGE, Arena, KVCache, KVPool, PagedSequence
from .config import ModelConfig, TextConfig


def _to_np(x):
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


class DecodeState:
    """Device-resident state of a batch of B decoding sequences (what one graph replay reads/writes)."""

    def __init__(self, lm: "LanguageModel", B: int, nsplit: int = 32, ring_len: int = 64):
        t, dev = lm.args, lm.device
        D, hd, Hq, Hkv = t.hidden_size, lm.head_dim, t.num_attention_heads, t.num_key_value_heads
        bf, i32 = torch.bfloat16, torch.int32
        self.B, self.nsplit, self.ring_len = B, nsplit, ring_len   # nsplit: buffers sized for the max, see decode_begin
        A = lm.arena.alloc
        self.tok = A(B, i32, zero=True)
        self.pos = A(B, i32, zero=True)
        self.ctx = A(B, i32, zero=True)
        self.step = A(1, i32, zero=True)
        self.h = A((B, D), bf)
        self.qkv = A((B, (Hq + 2 * Hkv) * hd), bf)
        self.attn = A((B, Hq * hd), bf)
        # (16 rows even for a narrower state: a 5..16-row step hands the SwiGLU output to the down projection in the MFMA tile's
        #  layout [inter / 8][16][8] - vlm_decode_args.flags VLM_DECODE_ACT16, csrc/gemv_mfma_rows.hip)
        self.act = A((max(B, 16), t.intermediate_size), bf)
        self.part_ml = A((B, Hq, nsplit, 2), torch.float32)
        self.out_ring = A((ring_len, B), i32, zero=True)
        # rows at pitch VL = vocab rounded up to 8 (the engine's pitch: 16-byte aligned rows for any vocabulary, e.g. 32003)
        VL = (t.vocab_size + 7) & ~7
        self.logits = torch.empty(B, VL, dtype=bf, device=dev)[:, : t.vocab_size]
        self.logprobs = torch.empty(B, VL, dtype=bf, device=dev)[:, : t.vocab_size]
        self.scratch = torch.empty(B, VL, dtype=bf, device=dev)[:, : t.vocab_size]
        self.part_o = torch.empty(B, Hq, nsplit, hd, dtype=torch.float32, device=dev)
        self.sample_ws = ops.sample_workspace(B, dev)
        # history of fed tokens for the logits processors (penalties): ring per row + count
        from ...sample_utils import HIST_CAP
        self.hist = A((B, HIST_CAP), i32, zero=True)
        self.hist_len = A(B, i32, zero=True)
        self._pen = None            # (key, PenaltyArgs, device bias tensors) kept alive while a graph may use them
        self.graph_key = None

    MAX_ROW_BIAS = 64           # logit_bias entries per request in the per-row form

    def row_penalty_tables(self):
        """Device tables of the per-request processors of a continuous batch (vlm_penalty_args.row_params): fp32 [B, 8]
        parameters and [B, MAX_ROW_BIAS] bias lists, allocated on first use, all zero = every row passes through."""
        if getattr(self, "row_params", None) is None:
            dev = self.hist.device
            self.row_params = torch.zeros(self.B, 8, dtype=torch.float32, device=dev)
            self.row_bias_idx = torch.zeros(self.B, self.MAX_ROW_BIAS, dtype=torch.int32, device=dev)
            self.row_bias_val = torch.zeros(self.B, self.MAX_ROW_BIAS, dtype=torch.float32, device=dev)
            self._pen_rows = _lib.PenaltyArgs(self.hist.data_ptr(), self.hist_len.data_ptr(), self.hist.shape[1], 0.0, 0, 0.0, 0,
                                              0.0, 0, self.row_bias_idx.data_ptr(), self.row_bias_val.data_ptr(), 0,
                                              self.row_params.data_ptr(), self.MAX_ROW_BIAS)
        return self._pen_rows

    @classmethod
    def pack_row_penalties(cls, specs, hist_cap: int):
        """specs: one sample_utils.LogitsProcessors (or None) per row -> host arrays (params f32 [n, 8], bias_idx i32
        [n, MAX_ROW_BIAS], bias_val f32 [n, MAX_ROW_BIAS]) in the layout of vlm_penalty_args.row_params"""
        n = len(specs)
        params = np.zeros((n, 8), dtype=np.float32)
        bidx = np.zeros((n, cls.MAX_ROW_BIAS), dtype=np.int32)
        bval = np.zeros((n, cls.MAX_ROW_BIAS), dtype=np.float32)
        for r, sp in enumerate(specs):
            if not sp:
                continue
            bias = sp.logit_bias or {}
            if len(bias) > cls.MAX_ROW_BIAS:
                raise NotImplementedError(f"more than {cls.MAX_ROW_BIAS} logit_bias entries per request in a continuous batch")
            params[r, :6] = (sp.repetition_penalty, min(sp.repetition_context_size, hist_cap), sp.presence_penalty,
                             min(sp.presence_context_size, hist_cap), sp.frequency_penalty, min(sp.frequency_context_size, hist_cap))
            params[r, 6] = len(bias)
            bidx[r, :len(bias)] = list(bias.keys())
            bval[r, :len(bias)] = list(bias.values())
        return params, bidx, bval

    def penalty_args(self, spec):
        """_lib.PenaltyArgs over this state's history for a sample_utils.LogitsProcessors spec (cached per spec);
        spec == "rows": the per-row tables of a continuous batch (row_penalty_tables)."""
        if isinstance(spec, str) and spec == "rows":
            return self.row_penalty_tables()
        if not spec:
            return None
        if self._pen is None or self._pen[0] != spec.key():
            bias = spec.logit_bias or {}
            idx = _lib.h2d(np.asarray(list(bias.keys()), dtype=np.int32), self.hist.device) if bias else None
            val = _lib.h2d(np.asarray(list(bias.values()), dtype=np.float32), self.hist.device) if bias else None
            pa = _lib.PenaltyArgs(self.hist.data_ptr(), self.hist_len.data_ptr(), self.hist.shape[1],
                                  float(spec.repetition_penalty), int(spec.repetition_context_size),
                                  float(spec.presence_penalty), int(spec.presence_context_size),
                                  float(spec.frequency_penalty), int(spec.frequency_context_size),
                                  idx.data_ptr() if idx is not None else None, val.data_ptr() if val is not None else None,
                                  len(bias))
            self._pen = (spec.key(), pa, idx, val)
        return self._pen[1]

    def set_history(self, rows_tokens):
        """rows_tokens[b] = the tokens fed so far for row b (the prompt): the last HIST_CAP of them go into the ring"""
        cap = self.hist.shape[1]
        host = np.zeros((len(rows_tokens), cap), dtype=np.int32)
        lens = np.zeros(len(rows_tokens), dtype=np.int32)
        for b, t in enumerate(rows_tokens):
            t = np.asarray(t, dtype=np.int64).reshape(-1)[-cap:]
            host[b, :len(t)] = t
            lens[b] = len(t)
        self.hist[:len(rows_tokens)].copy_(_lib.h2d(host, self.hist.device))
        self.hist_len[:len(rows_tokens)].copy_(_lib.h2d(lens, self.hist.device))

    def args(self, temperature=0.0, top_p=1.0, min_p=0.0, top_k=0, seed=0, with_logprobs=True, B=None, flags=0,
             penalties=None):
        """B < self.B: the step runs over the first B rows (every buffer is row-major, a prefix is a valid state).
        flags: _lib.DECODE_FUSED_TAIL = the step starts from h == embed[tok] and leaves the next step's h behind."""
        p = lambda t: t.data_ptr()  # noqa: E731
        return _lib.DecodeArgs(self.B if B is None else int(B), p(self.tok), p(self.pos), p(self.ctx), p(self.step), p(self.h), p(self.qkv),
                               p(self.attn), p(self.act), p(self.logits),
                               p(self.logprobs) if (with_logprobs or temperature > 0) else None, p(self.scratch),
                               p(self.part_o), p(self.part_ml), p(self.sample_ws), p(self.out_ring), self.ring_len,
                               self.nsplit, float(temperature), float(top_p), float(min_p), int(top_k),
                               int(seed) & 0xFFFFFFFF, int(flags) | _lib.DECODE_ACT16,       # (self.act holds 16 rows)
                               C.pointer(pa) if (pa := self.penalty_args(penalties)) is not None else None)


class LanguageModel:
    # rows of one decode step (decode_step_rows): up to 16 on the GEMV family (v_dot2c: 1 / 2 / 4 / 8, skinny-M MFMA: 5..16), 17..64 as
    # WIDE steps on the prefill GEMMs (needs the caller's block table; not for two-table RoPE models: phi3_v sets 16)
    MAX_DECODE_ROWS = 64

    supports_logits_to_keep = True

    def __init__(self, args: TextConfig, config: ModelConfig, device="cuda", kv_pool_tokens: int = 32768,
                 max_seqs: int = 64, kv_layout: str = "auto"):
        self.args = args
        self.config = config
        self.model_type = args.model_type
        self.device = device
        self._rope_deltas = None
        self._position_ids = None
        self._handle = None
        self._w: Dict[str, torch.Tensor] = {}
        self._kv_pool_tokens, self._max_seqs = kv_pool_tokens, max_seqs
        self._kv_layout = os.environ.get("VLM_KV_LAYOUT", kv_layout)   # env: A/B knob for measurements
        self.pool: Optional[KVPool] = None
        self._decode_states: Dict[int, DecodeState] = {}
        sec = list((args.rope_scaling or {}).get("mrope_section", [16, 24, 24]))
        self.mrope_section = sec

    # ------------------------------------------------------------------ properties (reference language.py:520-530)
    @property
    def layers(self):
        return list(range(self.args.num_hidden_layers))

    @property
    def head_dim(self):
        """An explicit `args.head_dim` wins (a model whose heads are zero-padded to a width the kernels have)."""
        return getattr(self.args, "head_dim", None) or self.args.hidden_size // self.args.num_attention_heads

    @property
    def n_kv_heads(self):
        return self.args.num_key_value_heads

    # ------------------------------------------------------------------ weights
    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: names relative to `language_model.` (`model.layers.i....`, `model.embed_tokens.weight`, `lm_head.weight`).
        Packs q/k/v into one [Hq*D + 2*Hkv*D, hidden] matrix and interleaves gate/up rows (g0,u0,g1,u1,...) so the
        SwiGLU product is an epilogue of one GEMM/GEMV."""
        t, dev = self.args, self.device
        L = _lib.lib()
        bf = torch.bfloat16

        from .. import quantized as Qz

        def g(name):
            return W[name].to(device=dev, dtype=bf)

        def lin(path):
            """a projection weight: bf16 tensor, or QuantW when the checkpoint holds `<path>.scales` (utils.py:961)"""
            if isinstance(W.get(path + ".weight"), Qz.QuantW):      # already packed by a model-specific loader (phi3_v)
                return W[path + ".weight"].to(dev)
            if Qz.has_scales(W, path):
                return Qz.take(W, path).to(dev)
            return g(path + ".weight")

        # small tensors (norm weights, biases, rope table, decode state, block table) live in ONE arena
        small_bytes = t.num_hidden_layers * (2 * t.hidden_size + (t.num_attention_heads + 2 * t.num_key_value_heads) * self.head_dim) * 2
        self.arena = Arena(small_bytes + (8 << 20), device=dev)
        # ... and the big matrices live in ONE allocation too, in the order a decode step streams them, so the
        # driver can map the region with large page fragments (A/B knob: VLM_WEIGHT_ARENA=0 -> separate tensors)
        qkv_rows = (t.num_attention_heads + 2 * t.num_key_value_heads) * self.head_dim
        per_layer = (qkv_rows + t.num_attention_heads * self.head_dim + 3 * t.intermediate_size) * t.hidden_size * 2
        n_big = t.num_hidden_layers * per_layer + (1 if t.tie_word_embeddings else 2) * (t.vocab_size + 8) * t.hidden_size * 2
        use_wa = os.environ.get("VLM_WEIGHT_ARENA", "1") != "0" and not any(k.endswith(".scales") or isinstance(v, Qz.QuantW) for k, v in W.items())
        self.warena = Arena(n_big + (4 * t.num_hidden_layers + 4) * 4096, device=dev, zero=False) if use_wa else None

        def big(x):
            if isinstance(x, Qz.QuantW):
                return x                       # 4-bit matrices keep their own (small) allocations
            x = x.contiguous()
            if self.warena is None:
                return x
            out = self.warena.alloc(x.shape, x.dtype, align=4096)
            out.copy_(x)
            return out

        cfg = _lib.LlmConfig(t.hidden_size, t.num_hidden_layers, t.intermediate_size, t.num_attention_heads,
                             t.num_key_value_heads, self.head_dim, t.vocab_size, float(t.rms_norm_eps),
                             int(self.mrope_section[0]), int(self.mrope_section[1]),
                             float(getattr(t, "attn_scale", 0.0) or 0.0), float(getattr(t, "rope_qk_scale", 0.0) or 0.0),
                             int(getattr(self, "rope_long_from", 0) or 0))
        h = C.c_void_p()
        check(L.vlm_llm_create(C.byref(cfg), C.byref(h)), "llm_create")
        self._handle = h
        self.apply_tuning()
        for i in range(t.num_hidden_layers):
            p = f"model.layers.{i}."
            qkv_parts = [lin(p + "self_attn.q_proj"), lin(p + "self_attn.k_proj"), lin(p + "self_attn.v_proj")]
            if any(isinstance(x, Qz.QuantW) for x in qkv_parts):
                if not all(isinstance(x, Qz.QuantW) for x in qkv_parts):
                    raise NotImplementedError("q / k / v projections of one layer must be all quantized or all bf16")
                wqkv = Qz.cat_rows(qkv_parts)
            else:
                wqkv = big(torch.cat(qkv_parts, dim=0))
            bqkv = torch.cat([g(p + "self_attn.q_proj.bias"), g(p + "self_attn.k_proj.bias"),
                              g(p + "self_attn.v_proj.bias")], dim=0).contiguous()
            gate, up = lin(p + "mlp.gate_proj"), lin(p + "mlp.up_proj")
            wo = big(lin(p + "self_attn.o_proj"))
            if isinstance(gate, Qz.QuantW) != isinstance(up, Qz.QuantW):
                raise NotImplementedError("gate / up projections of one layer must be both quantized or both bf16")
            wgu = Qz.interleave_rows(gate, up) if isinstance(gate, Qz.QuantW) else \
                big(torch.stack([gate, up], dim=1).reshape(2 * t.intermediate_size, t.hidden_size))
            del gate, up
            bqkv = self.arena.put(bqkv)
            ws = dict(ln1=self.arena.put(g(p + "input_layernorm.weight")), wqkv=wqkv, bqkv=bqkv, wo=wo,
                      ln2=self.arena.put(g(p + "post_attention_layernorm.weight")), wgu=wgu,
                      wdown=big(lin(p + "mlp.down_proj")))
            for k, v in ws.items():
                self._w[f"{i}.{k}"] = v

            def wp(x):
                return x.wq.data_ptr() if isinstance(x, Qz.QuantW) else x.data_ptr()

            def sp(x):
                return x.sb.data_ptr() if isinstance(x, Qz.QuantW) else None

            lay = _lib.LlmLayer(ws["ln1"].data_ptr(), wp(wqkv), bqkv.data_ptr(), wp(ws["wo"]), ws["ln2"].data_ptr(), wp(wgu),
                                wp(ws["wdown"]), sp(wqkv), sp(ws["wo"]), sp(wgu), sp(ws["wdown"]))
            check(L.vlm_llm_set_layer(h, i, C.byref(lay)), "llm_set_layer")
        def pad_rows8(x):
            """the head (and a tied embedding) gets zero rows up to a multiple of 8: the prefill logits GEMM runs over them"""
            n = (-x.shape[0]) % 8
            if n == 0:
                return x
            if isinstance(x, Qz.QuantW):
                return Qz.QuantW(torch.cat([x.wq, torch.zeros(n, x.wq.shape[1], dtype=x.wq.dtype, device=x.wq.device)]).contiguous(),
                                 torch.cat([x.sb, torch.zeros(n, x.sb.shape[1], dtype=x.sb.dtype, device=x.sb.device)]).contiguous())
            return torch.cat([x, torch.zeros(n, x.shape[1], dtype=x.dtype, device=x.device)])

        embed = big(pad_rows8(lin("model.embed_tokens")) if t.tie_word_embeddings else lin("model.embed_tokens"))
        head = embed if t.tie_word_embeddings else big(pad_rows8(lin("lm_head")))
        self.quantized = any(isinstance(v, Qz.QuantW) for v in self._w.values()) or isinstance(embed, Qz.QuantW) \
            or isinstance(head, Qz.QuantW)
        norm = self.arena.put(g("model.norm.weight"))
        hd = self.head_dim
        # compute_inv_freq (reference rope_utils.py:1042-1043), fp32 on the host
        # `args.rope_dim` < head_dim: only the first rope_dim / 2 pairs rotate (the rest get angle 0 = identity)
        rd = getattr(t, "rope_dim", None) or hd
        inv = torch.zeros(hd // 2, dtype=torch.float32)
        inv[: rd // 2] = 1.0 / (t.rope_theta ** (torch.arange(0, rd, 2).to(torch.float32) / rd))
        if getattr(t, "inv_freq", None) is not None:      # a model's own table (SuScaledRoPE: 1 / (factor * theta ** (2 i / d)))
            own = torch.as_tensor(t.inv_freq, dtype=torch.float32).reshape(-1)
            inv.zero_()
            inv[: own.numel()] = own
        if getattr(t, "inv_freq_long", None) is not None:   # second table behind the first (vlm_llm_config.rope_long_from)
            own = torch.as_tensor(t.inv_freq_long, dtype=torch.float32).reshape(-1)
            inv = torch.cat([inv, torch.zeros(hd // 2, dtype=torch.float32)])
            inv[hd // 2: hd // 2 + own.numel()] = own
        inv_freq = self.arena.put(inv.to(dev))
        self._w.update(embed=embed, head=head, norm=norm, inv_freq=inv_freq)
        isq = lambda x: isinstance(x, Qz.QuantW)    # noqa: E731
        gl = _lib.LlmGlobals(embed.wq.data_ptr() if isq(embed) else embed.data_ptr(), norm.data_ptr(),
                             head.wq.data_ptr() if isq(head) else head.data_ptr(), inv_freq.data_ptr(),
                             embed.sb.data_ptr() if isq(embed) else None, head.sb.data_ptr() if isq(head) else None)
        check(L.vlm_llm_set_globals(h, C.byref(gl)), "llm_set_globals")
        self._init_pool()

    # decode-step tuning (results are identical under every setting up to bf16 ties; defaults from the measurements in
    # DESIGN.md, environment overrides VLM_DECODE_<NAME> for A/B runs)
    TUNING_DEFAULTS = {"fused_tail": 1, "mfma_gemv": 1, "attn_pagesplit": 16, "gemv_variant": 1, "attn_merge": 1}

    def apply_tuning(self, **over):
        L = _lib.lib()
        t = dict(self.TUNING_DEFAULTS)
        for k in t:
            env = os.environ.get("VLM_DECODE_" + k.upper())
            if env is not None:
                t[k] = int(env, 0)
        t.update(over)
        self.tuning = t
        for key, name in ((_lib.TUNE_MFMA_GEMV, "mfma_gemv"), (_lib.TUNE_ATTN_PAGESPLIT, "attn_pagesplit"),
                          (_lib.TUNE_GEMV_VARIANT, "gemv_variant"), (_lib.TUNE_ATTN_MERGE, "attn_merge")):
            check(L.vlm_llm_set_tuning(self._handle, key, int(t[name])), "llm_set_tuning")
        for st in getattr(self, "_decode_states", {}).values():
            st.graph_key = None          # the engine dropped its captured steps

    def _init_pool(self):
        t = self.args
        self.pool = KVPool(t.num_hidden_layers, t.num_key_value_heads, self.head_dim, self._kv_pool_tokens,
                           self._max_seqs, max_pages_per_seq=min(512, (self._kv_pool_tokens + PAGE - 1) // PAGE),
                           device=self.device, arena=self.arena, layout=self._kv_layout)
        kv = _lib.KvPool(self.pool.kpool.data_ptr(), self.pool.vpool.data_ptr(), self.pool.layer_stride,
                         self.pool.block_table.data_ptr(), self.pool.max_pages)
        check(_lib.lib().vlm_llm_set_kv(self._handle, C.byref(kv)), "llm_set_kv")

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().vlm_llm_destroy(self._handle)
        except Exception:
            pass

    @property
    def embed_tokens_weight(self):
        return self._w["embed"]

    def embed_tokens(self, input_ids) -> torch.Tensor:
        """nn.Embedding (reference language.py:164,179).  input_ids [B, L] -> [B, L, D]"""
        ids = _lib.h2d(np.asarray(_to_np(input_ids), dtype=np.int32), self.device)
        B, Lq = ids.shape
        e = self._w["embed"]
        if hasattr(e, "wq"):          # nn.QuantizedEmbedding: dequantize of the looked-up rows
            out = ops.dequant_w4(e.wq, e.sb, rows=ids.reshape(-1).contiguous())
        else:
            out = ops.embed_gather(ids.reshape(-1), e)
        return out.view(B, Lq, -1)

    # ------------------------------------------------------------------ cache
    def make_cache(self) -> List[KVCache]:
        seq = PagedSequence(self.pool)
        return [KVCache(seq, i) for i in range(self.args.num_hidden_layers)]

    def make_cache_batch(self, n: int) -> List[List[KVCache]]:
        """n caches on consecutive block-table rows (what a decode batch needs)."""
        rows = self.pool.new_seqs(n)
        return [[KVCache(s, i) for i in range(self.args.num_hidden_layers)]
                for s in (PagedSequence(self.pool, r) for r in rows)]

    # ------------------------------------------------------------------ get_rope_index (reference language.py:216-402)
    def get_rope_index(self, input_ids, image_grid_thw=None, video_grid_thw=None, attention_mask=None):
        cfg = self.config
        input_ids = _to_np(input_ids)
        image_grid_thw, video_grid_thw = _to_np(image_grid_thw), _to_np(video_grid_thw)
        attention_mask = _to_np(attention_mask)
        B, L = input_ids.shape
        ms = cfg.vision_config.spatial_merge_size
        img_id, vid_id, vstart = cfg.image_token_id, cfg.video_token_id, cfg.vision_start_token_id
        if image_grid_thw is not None or video_grid_thw is not None:
            if attention_mask is None:
                attention_mask = np.ones_like(input_ids)
            position_ids = np.ones((3, B, L), dtype=np.int64)
            deltas = []
            ii = vi = 0
            for i in range(B):
                row_mask = attention_mask[i]
                toks = input_ids[i][row_mask == 1]
                tl = toks.tolist()
                starts = np.nonzero(toks[:-1] == vstart)[0]
                vision_tokens = toks[starts + 1]
                image_nums = int((vision_tokens == img_id).sum())
                video_nums = int((vision_tokens == vid_id).sum())
                chunks: List[np.ndarray] = []
                st, cur_max = 0, -1
                ri, rv = image_nums, video_nums
                for _ in range(image_nums + video_nums):
                    ed_image = tl.index(img_id, st) if (img_id in tl and ri > 0) else len(tl) + 1
                    ed_video = tl.index(vid_id, st) if (vid_id in tl and rv > 0) else len(tl) + 1
                    if ed_image < ed_video:
                        t_, h_, w_ = [int(x) for x in image_grid_thw[ii]]
                        ii += 1
                        ri -= 1
                        ed = ed_image
                    else:
                        t_, h_, w_ = [int(x) for x in video_grid_thw[vi]]
                        vi += 1
                        rv -= 1
                        ed = ed_video
                    gt, gh, gw = t_, h_ // ms, w_ // ms
                    text_len = ed - st
                    st_idx = cur_max + 1
                    chunks.append(np.broadcast_to(np.arange(text_len)[None], (3, text_len)) + st_idx)
                    tix = np.broadcast_to(np.arange(gt)[:, None], (gt, gh * gw)).reshape(-1)
                    hix = np.broadcast_to(np.arange(gh)[None, :, None], (gt, gh, gw)).reshape(-1)
                    wix = np.broadcast_to(np.arange(gw)[None, None, :], (gt, gh, gw)).reshape(-1)
                    vis = np.stack([tix, hix, wix]) + text_len + st_idx
                    chunks.append(vis)
                    cur_max = int(vis.max())   # == llm_pos_ids_list[-1].max() in the reference
                    st = ed + gt * gh * gw
                if st < len(tl):
                    st_idx = cur_max + 1
                    text_len = len(tl) - st
                    chunks.append(np.broadcast_to(np.arange(text_len)[None], (3, text_len)) + st_idx)
                if not chunks:
                    deltas.append(0)
                    continue
                llm_positions = np.concatenate(chunks, axis=1).reshape(3, -1)
                position_ids[:, i, row_mask == 1] = llm_positions
                deltas.append(int(llm_positions.max()) + 1 - len(tl))
            return position_ids, np.array(deltas, dtype=np.int64).reshape(-1, 1)
        if attention_mask is not None:
            am = attention_mask.astype(np.int64)
            position_ids = np.cumsum(am, axis=-1) - 1
            position_ids = np.where(am == 0, 1, position_ids)
            deltas = position_ids.max(axis=-1, keepdims=True) + 1 - am.shape[-1]
            return position_ids, deltas
        position_ids = np.broadcast_to(np.arange(L)[None], (B, L)).astype(np.int64)
        return position_ids, np.zeros((B, 1), dtype=np.int64)

    # ------------------------------------------------------------------ prefill over concatenated sequences
    def prefill(self, inputs_embeds: torch.Tensor, position_ids: np.ndarray, caches: List[List[KVCache]],
                lengths: List[int], logits_rows: str = "last", reserve_extra: int = 0) -> torch.Tensor:
        """inputs_embeds [T, D] (sequences concatenated, lengths[i] tokens each); position_ids int [3, T];
        caches[i] = the layer views of sequence i (appended after their current offset).
        -> logits [n_seq, V] (logits_rows == "last") or [T, V] ("all")."""
        t, dev = self.args, self.device
        T, D = inputs_embeds.shape
        hd, Hq, Hkv = self.head_dim, t.num_attention_heads, t.num_key_value_heads
        seqs = [c[0]._seq for c in caches]
        if any(s.offset != 0 for s in seqs):
            return self._prefill_onto_cache(inputs_embeds, position_ids, caches, lengths, logits_rows, reserve_extra)
        cu = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
        kv_seq = np.concatenate([np.full(n, s.seq, dtype=np.int32) for n, s in zip(lengths, seqs)])
        kv_slot = np.concatenate([np.arange(n, dtype=np.int32) for n in lengths])
        for n, s in zip(lengths, seqs):
            if s.rotating and n > s.max_size and self.ROTATING_PROMPT_WINDOW_MASK:
                # the reference's Phi-3.5-V hands cache[0] to create_attention_mask (phi3_v.py:163): for a first prompt longer
                # than max_kv_size RotatingKVCache.make_mask returns a causal mask WINDOWED to max_size (cache.py:586-596);
                # Qwen2-VL and Bunny pass the cache list and attend causally.  The windowed prefill is not built
                raise NotImplementedError(f"max_kv_size={s.max_size} with a prompt of {n} tokens: this family's reference prefills "
                                          "such a prompt under a sliding-window mask (RotatingKVCache.make_mask); only prompts "
                                          "up to max_kv_size are built for it")
            # (a rotating window never holds more than max(prompt, max_size) + 1 entries: max_kv_size bounds the reservation)
            s.reserve(max(n, s.max_size) + 1 if s.rotating else n + reserve_extra)
        kv = self._kv_struct(0)   # prefill addresses block-table rows by absolute sequence id
        check(_lib.lib().vlm_llm_set_kv(self._handle, C.byref(kv)), "llm_set_kv")
        nqb = int(sum((n + 127) // 128 for n in lengths))
        pos = np.ascontiguousarray(position_ids, dtype=np.int32)
        i32 = torch.int32
        pos_d = _lib.h2d(pos, dev)
        meta = _lib.h2d(np.concatenate([kv_seq, kv_slot, cu]), dev)
        kv_seq_d, kv_slot_d, cu_d = meta[:T], meta[T:2 * T], meta[2 * T:]
        if logits_rows == "last":
            rows = (cu[1:] - 1).astype(np.int32)
        else:
            rows = np.arange(T, dtype=np.int32)
        rows_d = _lib.h2d(rows, dev)
        bf = torch.bfloat16
        h = inputs_embeds.contiguous()
        xn = torch.empty(T, D, dtype=bf, device=dev)
        qkv = torch.empty(T, (Hq + 2 * Hkv) * hd, dtype=bf, device=dev)
        attn = torch.empty(T, Hq * hd, dtype=bf, device=dev)
        act = torch.empty(T, t.intermediate_size, dtype=bf, device=dev)
        xlast = torch.empty(len(rows), D, dtype=bf, device=dev)
        logits = torch.empty(len(rows), (t.vocab_size + 7) & ~7, dtype=bf, device=dev)[:, : t.vocab_size]      # pitch: see DecodeState
        a = _lib.PrefillArgs(h.data_ptr(), T, pos_d[0].data_ptr(), pos_d[1].data_ptr(), pos_d[2].data_ptr(),
                             kv_seq_d.data_ptr(), kv_slot_d.data_ptr(), cu_d.data_ptr(), len(lengths), nqb,
                             xn.data_ptr(), qkv.data_ptr(), attn.data_ptr(), act.data_ptr(), rows_d.data_ptr(),
                             len(rows), xlast.data_ptr(), logits.data_ptr(), int(bool(getattr(self, "_prefill_rope_long", False))))
        check(_lib.lib().vlm_llm_prefill(self._handle, C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "llm_prefill")
        for n, s in zip(lengths, seqs):
            s.offset += n
            s.note_prefill(n)
        self._keep = (h, xn, qkv, attn, act, xlast, pos_d, meta, rows_d)  # keep alive until the stream has run
        return logits

    def _prefill_onto_cache(self, inputs_embeds, position_ids, caches, lengths, logits_rows, reserve_extra):
        """A prompt chunk appended to a NON-EMPTY cache: chunked prefill (reference ar.py:426-472) and `prompt_cache=`
        continuation across calls, i.e. multi-turn (dispatch.py:861-882, common.py:243-263).  The chunk's queries attend
        to [cached tokens | the chunk] (cache.py:345-367 + base.py:366-373 with the causal mask offset by the cache
        length).  Kept simple: the layer loop runs here over the C-ABI operators; per layer the cached k / v rows are fetched
        back into a full-length token-major buffer (vlm_kv_gather) and the varlen causal attention runs with the segment's
        query rows starting at the cache length (vlm_attn_prefill's q_start, round 6: the prefix rows are keys only - they used
        to carry zero queries, O(Tf^2) per chunk)."""
        if any(c[0]._seq.rotating for c in caches):
            raise NotImplementedError("max_kv_size: a multi-token update of a non-empty rotating cache (the reference first trims "
                                      "the window to max_size - 1 + S, cache.py:486-505) is not built; only the first prompt")
        t, dev = self.args, self.device
        hd, Hq, Hkv = self.head_dim, t.num_attention_heads, t.num_key_value_heads
        D, QKV = t.hidden_size, (Hq + 2 * Hkv) * self.head_dim
        seqs = [c[0]._seq for c in caches]
        if any(getattr(s, "q8", False) for s in seqs):
            raise NotImplementedError("a prompt chunk onto a QUANTIZED KV cache (kv_bits): only decode steps attend over the "
                                      "8-bit pools; continue such a conversation without kv_bits or from a fresh cache")
        offs = [int(s.offset) for s in seqs]
        for n, s in zip(lengths, seqs):
            s.reserve(s.offset + n + reserve_extra)
        pool = self.pool
        bt = pool.block_table
        bf, i32 = torch.bfloat16, torch.int32
        T = int(sum(lengths))
        tot = [o + n for o, n in zip(offs, lengths)]
        cu_full = np.concatenate([[0], np.cumsum(tot)]).astype(np.int32)
        Tf = int(cu_full[-1])
        # rows of the full-length buffer: per sequence [prefix | chunk]
        new_rows = np.concatenate([np.arange(cu_full[i] + offs[i], cu_full[i + 1]) for i in range(len(seqs))]).astype(np.int64)
        old_rows = np.concatenate([np.arange(cu_full[i], cu_full[i] + offs[i]) for i in range(len(seqs))]).astype(np.int64)
        full_seq = np.concatenate([np.full(tot[i], seqs[i].seq, np.int32) for i in range(len(seqs))])
        full_slot = np.concatenate([np.arange(tot[i], dtype=np.int32) for i in range(len(seqs))])
        pos = np.ascontiguousarray(position_ids, dtype=np.int32)
        pos_d = _lib.h2d(pos, dev)
        new_rows_d, old_rows_d = _lib.h2d(new_rows, dev), _lib.h2d(old_rows, dev)
        seq_d, slot_d, cu_d = _lib.h2d(full_seq, dev), _lib.h2d(full_slot, dev), _lib.h2d(cu_full, dev)
        new_seq_d, new_slot_d = seq_d[new_rows_d], slot_d[new_rows_d]
        old_seq_d, old_slot_d = seq_d[old_rows_d].contiguous(), slot_d[old_rows_d].contiguous()
        nqb = int(sum((n + 127) // 128 for n in lengths))      # query blocks: the chunk rows only
        qstart_d = _lib.h2d(np.asarray(offs, dtype=np.int32), dev)
        if os.environ.get("VLM_ONTO_CACHE_QSTART") == "0":     # A/B knob: the form of rounds 3-5 (the prefix rows carry zero queries)
            nqb, qstart_d = int(sum((n + 127) // 128 for n in tot)), None
        scale = float(getattr(t, "attn_scale", 0.0) or 0.0) or hd ** -0.5
        # A SHORT chunk (a conversation turn of up to 64 tokens in all) takes the paged DECODE attention instead: every new token is
        # a decode row over its sequence's pages with kv_len = cache length + its index + 1 (its K / V are in the pages by then:
        # mrope_kvwrite_ below) - no gather of the prefix at all, and the keys are split over workgroups as in a decode step
        # (one 64-row query block per head walking 16k keys took ~0.4 ms per layer; profiles/r06_long_prompt_prefill.txt).
        short = T <= 64 and os.environ.get("VLM_ONTO_CACHE_DECODE_ATTN") != "0"
        if short:
            row_seq = np.concatenate([np.full(n, s.seq, np.int64) for n, s in zip(lengths, seqs)])
            row_len = np.concatenate([o + 1 + np.arange(n) for o, n in zip(offs, lengths)]).astype(np.int32)
            bt_rows = bt[_lib.h2d(row_seq, dev)].contiguous()
            row_len_d = _lib.h2d(row_len, dev)
            longest = int(row_len.max())
            dec_nsplit = 1 if longest <= 2048 else max(2, min(32, (longest + 16 * PAGE - 1) // (16 * PAGE)))
        h = inputs_embeds.contiguous().clone()
        sec = self.mrope_section
        def dense(x):        # 4-bit matrices are materialised as bf16 for the GEMMs of this (rare) path
            return ops.dequant_w4(x.wq, x.sb) if hasattr(x, "wq") else x

        for i in range(t.num_hidden_layers):
            # only THIS layer's matrices are dequantised here (the head once, after the loop)
            w = {k: (dense(v) if k.startswith(f"{i}.") else v) for k, v in self._w.items()
                 if k.startswith(f"{i}.") or k in ("inv_freq", "norm")} if self.quantized else self._w
            kp, vp = pool.kpool[i], pool.vpool[i]          # this layer's K / V pools (flat views)
            xn = ops.rmsnorm(h, w[f"{i}.ln1"], t.rms_norm_eps)
            qkv = ops.gemm(xn, w[f"{i}.wqkv"], bias=w[f"{i}.bqkv"], epilogue=ops.EPI_BIAS)
            inv_tab = w["inv_freq"][hd // 2:] if getattr(self, "_prefill_rope_long", False) else w["inv_freq"]   # SuScaledRoPE regime of this call
            ops.mrope_kvwrite_(qkv, Hq, Hkv, hd, pos_d[0], pos_d[1], pos_d[2], inv_tab, int(sec[0]), int(sec[1]),
                               kv_seq=new_seq_d.contiguous(), kv_slot=new_slot_d.contiguous(), block_table=bt, kpool=kp, vpool=vp,
                               qk_scale=getattr(t, "rope_qk_scale", None))
            if short:
                attn = ops.attn_decode_paged(qkv, kp, vp, bt_rows, row_len_d, 0, Hq, Hkv, hd, scale, dec_nsplit)
                h = ops.gemm(attn, w[f"{i}.wo"], res=h, epilogue=ops.EPI_RESIDUAL)
                xn = ops.rmsnorm(h, w[f"{i}.ln2"], t.rms_norm_eps)
                act = ops.gemm(xn, w[f"{i}.wgu"], epilogue=ops.EPI_SWIGLU)
                h = ops.gemm(act, w[f"{i}.wdown"], res=h, epilogue=ops.EPI_RESIDUAL)
                continue
            full = (torch.empty if qstart_d is not None else torch.zeros)(Tf, QKV, dtype=bf, device=dev)       # (q_start: the q columns of the prefix rows are never read)
            full[new_rows_d] = qkv
            if old_rows.size:
                # ONE gather over every row of the buffer, straight into it: the chunk's rows re-read from the pages what
                # mrope_kvwrite_ just stored there (the same bits) - no prefix buffer, no second copy of Tf rows
                ops.kv_gather_(full, Hq, Hkv, hd, slot_d, bt, kp, vp, kv_seq=seq_d)
            attn = ops.attn_prefill(full, full[:, Hq * hd:], full[:, (Hq + Hkv) * hd:], cu_d, nqb, Hq, Hkv, hd, scale, True,
                                    q_start=qstart_d)
            h = ops.gemm(attn[new_rows_d].contiguous(), w[f"{i}.wo"], res=h, epilogue=ops.EPI_RESIDUAL)
            xn = ops.rmsnorm(h, w[f"{i}.ln2"], t.rms_norm_eps)
            act = ops.gemm(xn, w[f"{i}.wgu"], epilogue=ops.EPI_SWIGLU)
            h = ops.gemm(act, w[f"{i}.wdown"], res=h, epilogue=ops.EPI_RESIDUAL)
        cu_new = np.concatenate([[0], np.cumsum(lengths)])
        rows = (cu_new[1:] - 1) if logits_rows == "last" else np.arange(T)
        xl = ops.rmsnorm(h[_lib.h2d(rows.astype(np.int64), dev)].contiguous(), self._w["norm"], t.rms_norm_eps)
        logits = ops.gemm(xl, dense(self._w["head"]))[:, : t.vocab_size]
        for n, s in zip(lengths, seqs):
            s.offset += n
        return logits

    # ------------------------------------------------------------------ fused_greedy_decode (reference hook)
    def supports_fused_greedy_logits_processors(self, processors) -> bool:
        """reference ar.py:1023-1031: Python logits processors cannot run inside the captured step"""
        return False

    def fused_greedy_decode(self, inputs, cache=None, **kwargs):
        """The reference's plug point for a native greedy decode step (generate/ar.py:1015-1042: GenerationBatch calls
        `language_model.fused_greedy_decode(inputs[:, None], cache=prompt_cache, **fwd_kwargs)` and falls back to the
        module call when it returns None): feed one token per row, run the whole step on the device - layers, lm_head,
        log-softmax, argmax, cache / position advance - and return the sampled token ids [B] without a host round trip.
        Returns None (= fall back) for anything the captured step does not cover."""
        if cache is None or kwargs.get("logits_processors"):
            return None
        ids = _to_np(inputs).reshape(-1)
        B = int(ids.size)
        caches = [cache] if isinstance(cache[0], KVCache) else cache
        if B not in (1, 2, 4, 8) or len(caches) != B or any(c[0].offset == 0 for c in caches):
            return None
        if any(c[0]._seq.rotating for c in caches):       # max_kv_size: the host makes room before every step (the module call)
            return None
        deltas = self._rope_deltas if self._rope_deltas is not None else np.zeros((B, 1), dtype=np.int64)
        deltas = np.broadcast_to(np.asarray(deltas).reshape(-1, 1), (B, 1)) if np.asarray(deltas).size == 1 else np.asarray(deltas)
        try:
            st = self.decode_begin(caches, ids, deltas[:B], max_new_tokens=1)
        except RuntimeError:          # rows not on consecutive block-table rows: let the caller use the module call
            return None
        self.decode_run(st, 1, dict(temperature=0.0, top_p=1.0, min_p=0.0, top_k=0, seed=0))
        return st.tok[:B].clone()

    # ------------------------------------------------------------------ decode
    def decode_state(self, B: int) -> DecodeState:
        st = self._decode_states.get(B)
        if st is None:
            st = DecodeState(self, B)
            self._decode_states[B] = st
        return st

    # max_kv_size: the family's rule for the rope offset of a decode step over a RotatingKVCache.  The reference's Qwen2-VL reads
    # `cache[0]._idx` - the ring's WRITE INDEX - where the cache has one (qwen2_vl/language.py:426-431), so after the first wrap
    # a token's rope position is its ring index; the plain-rope families read `cache.offset` (llava_bunny/language.py:65-66,
    # idefics2/language.py:54-55, phi3_v/phi3_v.py:82-83)
    ROTATING_POS_FROM_RING = True
    # does the family's reference prefill a prompt longer than max_kv_size under the rotating cache's WINDOWED mask (it passes
    # cache[0] to create_attention_mask: phi3_v.py:163) instead of the plain causal one (cache list: Qwen2-VL, Bunny)?
    ROTATING_PROMPT_WINDOW_MASK = False

    def _rotate_windows(self, seqs) -> np.ndarray:
        """Before a one-token step over sequences with max_kv_size: read each one's rope offset (BEFORE the ring wraps, as the
        reference's forward does), then make room as RotatingKVCache._update_in_place would (cache.py:507-547) - the moves of
        PagedSequence.rotate_plan in one vlm_kv_move_tokens launch.  -> rope offsets int32 [B]"""
        if any(s.q8 for s in seqs if s.rotating):
            raise NotImplementedError("RotatingKVCache Quantization NYI")          # (the reference's own words, cache.py:583-584)
        rope_pos = np.array([(s.rope_offset if self.ROTATING_POS_FROM_RING else s.offset) for s in seqs], dtype=np.int32)
        rows, src, dst = [], [], []
        for s in seqs:
            plan = s.rotate_plan()
            if plan:
                rows += [s.seq] * len(plan[0])
                src += list(plan[0])
                dst += list(plan[1])
        if rows:
            pool = self.pool
            dev = _lib.h2d(np.array([rows, src, dst], dtype=np.int32), self.device)
            check(_lib.lib().vlm_kv_move_tokens(pool.kpool.data_ptr(), pool.vpool.data_ptr(), pool.layer_stride, pool.n_layers,
                                                dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), len(rows),
                                                pool.block_table.data_ptr(), pool.max_pages, pool.n_kv_heads, pool.head_dim,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "kv_move_tokens")
            self._keep_rot = dev
        return rope_pos

    def decode_begin(self, caches: List[List[KVCache]], first_tokens, rope_deltas, max_new_tokens: int, rope_pos=None) -> DecodeState:
        """Bind B sequences (B in {1,2,4,8}) to the device decode state: tok = first sampled tokens,
        ctx = cache offsets, pos = offset + rope_delta (reference language.py:476-509)."""
        B = len(caches)
        st = self.decode_state(B)
        seqs = [c[0]._seq for c in caches]
        for s in seqs:
            s.reserve(s.offset + max_new_tokens + 1)
        # attention decomposition for this generation: up to 2048 tokens one workgroup per (sequence, kv head)
        # walks the pages itself (nsplit = 1, no merge pass); beyond that, split-K with one workgroup per
        # page-stride (<= 32 splits) merged in the o_proj prologue
        max_total = max(s.kv_entries for s in seqs) + max_new_tokens + 1
        st.nsplit = 1 if max_total <= 2048 else max(2, min(32, (max_total + 16 * PAGE - 1) // (16 * PAGE)))
        if os.environ.get("VLM_DECODE_NSPLIT"):     # A/B knob for measurements
            st.nsplit = max(1, min(32, int(os.environ["VLM_DECODE_NSPLIT"])))
        # the engine indexes block-table rows by batch row: sequences must sit in rows 0..B-1 of a view
        rows = [s.seq for s in seqs]
        if rows != list(range(rows[0], rows[0] + B)):
            raise RuntimeError("decode batch needs consecutive KV sequence slots")
        st.seq_row0 = rows[0]
        # (a rotating window: entries held, not tokens seen; the rope offset is the family's - `rope_pos` when the caller read it
        # before the window wrapped, LanguageModel.__call__)
        ctx = np.array([s.kv_entries for s in seqs], dtype=np.int32)
        base = np.array([s.offset for s in seqs], dtype=np.int32) if rope_pos is None else np.asarray(rope_pos, dtype=np.int32)
        pos = base + np.asarray(rope_deltas, dtype=np.int64).reshape(-1).astype(np.int32)
        host = np.concatenate([pos, ctx, np.zeros(1, np.int32)])
        dev = _lib.h2d(host, self.device)
        st.pos.copy_(dev[:B]); st.ctx.copy_(dev[B:2 * B]); st.step.copy_(dev[2 * B:])
        if isinstance(first_tokens, torch.Tensor):
            st.tok.copy_(first_tokens.reshape(-1).to(torch.int32))
        else:
            st.tok.copy_(_lib.h2d(np.asarray(first_tokens, dtype=np.int32).reshape(-1), self.device))
        st.seqs = seqs
        # a fused-tail step starts from h == embed[tok] (vlm_decode_args.flags); every later h is left behind by the
        # previous step's sampler tail
        if not hasattr(self._w["embed"], "wq"):
            ops.embed_gather(st.tok[:B], self._w["embed"], out=st.h[:B])
        return st

    # ------------------------------------------------------------------ uniform 8-bit KV cache
    def quantize_kv(self, seqs, bits: int = 8, group_size: int = 64):
        """KVCache.to_quantized (reference cache.py:415-423) for whole sequences: every cached token of `seqs` (all layers) is
        quantised from the bf16 pools into the 8-bit pools (mx.quantize, group 64, 8 bits); from then on their decode steps
        attend over the 8-bit pools and quantise each new token (QuantizedKVCache.update_and_fetch, cache.py:233-334)."""
        if int(bits) != 8 or int(group_size) != 64:
            raise NotImplementedError(f"quantized KV cache: kv_bits = 8 with kv_group_size = 64 is built (asked: {bits} / {group_size})")
        seqs = [s for s in seqs if not s.q8]
        if not seqs:
            return
        pool = self.pool.ensure_q8()
        rows = np.concatenate([np.full(s.offset, s.seq, dtype=np.int32) for s in seqs]) if seqs else np.zeros(0, np.int32)
        slots = np.concatenate([np.arange(s.offset, dtype=np.int32) for s in seqs])
        if rows.size:
            dev = _lib.h2d(np.stack([rows, slots]), self.device)
            check(_lib.lib().vlm_kv_quantize_tokens(pool.kpool.data_ptr(), pool.vpool.data_ptr(), pool.kpool8.data_ptr(),
                                                    pool.vpool8.data_ptr(), pool.ksb.data_ptr(), pool.vsb.data_ptr(),
                                                    pool.layer_stride, pool.n_layers, dev[0].data_ptr(), dev[1].data_ptr(),
                                                    int(rows.size), pool.block_table.data_ptr(), pool.max_pages, pool.n_kv_heads,
                                                    pool.head_dim, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "kv_quantize_tokens")
            self._keep_q8 = dev
        for s in seqs:
            s.q8 = True

    def _kv_struct(self, row0: int, decode: bool = False, q8: bool = False):
        """The pool as the engine sees it for batch rows row0, row0+1, ...  Decode over an identity-layout pool:
        no block table (NULL) and pool pointers advanced to row0's region - the kernels compute
        page = row * max_pages + index instead of loading it."""
        pool = self.pool
        if decode and pool.identity:
            elems = row0 * pool.max_pages * pool.n_kv_heads * PAGE * pool.head_dim
            off = elems * pool.kpool.element_size()
            q = (pool.kpool8.data_ptr() + elems, pool.vpool8.data_ptr() + elems, pool.ksb.data_ptr() + elems // 64 * 4,
                 pool.vsb.data_ptr() + elems // 64 * 4) if q8 else (None, None, None, None)
            return _lib.KvPool(pool.kpool.data_ptr() + off, pool.vpool.data_ptr() + off, pool.layer_stride, None,
                               pool.max_pages, *q)
        bt = pool.block_table[row0:]
        q = (pool.kpool8.data_ptr(), pool.vpool8.data_ptr(), pool.ksb.data_ptr(), pool.vsb.data_ptr()) if q8 else (None,) * 4
        return _lib.KvPool(pool.kpool.data_ptr(), pool.vpool.data_ptr(), pool.layer_stride, bt.data_ptr(), pool.max_pages, *q)

    def decode_run(self, st: DecodeState, n_steps: int, sampler_args: dict, use_graph: bool = True, penalties=None):
        """Enqueue n_steps decode steps (graph replays when use_graph).  penalties: sample_utils.LogitsProcessors."""
        L = _lib.lib()
        if any(s.rotating for s in st.seqs):
            raise NotImplementedError("max_kv_size: the captured step writes at slot = tokens seen; a rotating window runs "
                                      "through the module call (LanguageModel.__call__), one step at a time")
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        q8 = self._seqs_q8(st.seqs)
        kv = self._kv_struct(st.seq_row0, decode=True, q8=q8)
        check(L.vlm_llm_set_kv(self._handle, C.byref(kv)), "llm_set_kv")
        # the fused tails (greedy: vlm_sample_greedy_advance; with a temperature: the sampler's last launch also advances and
        # gathers) read bf16 embedding rows
        fused = bool(self.tuning.get("fused_tail")) and not hasattr(self._w["embed"], "wq")
        args = st.args(flags=_lib.DECODE_FUSED_TAIL if fused else 0, penalties=penalties, **sampler_args)
        if os.environ.get("VLM_NO_GRAPH"):           # diagnostics: every launch of the step visible to the HIP runtime's log
            use_graph = False
        if use_graph:
            key = (st.seq_row0, st.nsplit, fused, q8, penalties.key() if penalties else None, tuple(sorted(sampler_args.items())))
            if st.graph_key != key or getattr(self, "_graph_owner", None) is not st:
                check(L.vlm_llm_decode_graph_build(self._handle, C.byref(args), stream), "decode_graph_build")
                st.graph_key = key
                self._graph_owner = st
            for _ in range(n_steps):
                check(L.vlm_llm_decode_graph_launch(self._handle, stream), "decode_graph_launch")
        else:
            for _ in range(n_steps):
                check(L.vlm_llm_decode_step(self._handle, C.byref(args), stream), "decode_step")
        for s in st.seqs:
            s.offset += n_steps

    @staticmethod
    def _seqs_q8(seqs) -> bool:
        flags = {bool(getattr(s, "q8", False)) for s in seqs}
        if len(flags) > 1:
            raise RuntimeError("a decode step cannot mix sequences with and without a quantized KV cache")
        return bool(flags) and flags.pop()

    def decode_step_rows(self, st: DecodeState, B: int, block_table: torch.Tensor, sampler_args: dict,
                         use_graph: bool = True, with_logprobs: bool = True, row_penalties: bool = False, q8: bool = False):
        """One decode step over rows 0..B-1 of `st` (B in {1, 2, 4, 8}), addressing the KV pool through the caller's own
        `block_table` (int32 [>= B, max_pages]).  A continuous batch keeps such a table so that a sequence changes
        batch row by copying one table row - no KV bytes move (the reference's `filter`/`extend` copy the caches,
        cache.py:1100-1201).  Graphs are cached per (B, nsplit, table, sampler) inside the engine."""
        L = _lib.lib()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        pool = self.pool
        q = (pool.kpool8.data_ptr(), pool.vpool8.data_ptr(), pool.ksb.data_ptr(), pool.vsb.data_ptr()) if q8 else (None,) * 4
        # the BATCH policy of the reference (models/cache.py:8-21, generate/ar.py:842-858): with kv_bits the last layer of a
        # stack deeper than 2 keeps its unquantised cache (vlm_kv_pool.q8_skip_last; the bf16 pools hold every token anyway)
        kv = _lib.KvPool(pool.kpool.data_ptr(), pool.vpool.data_ptr(), pool.layer_stride, block_table.data_ptr(),
                         pool.max_pages, *q, 1 if q8 else 0)
        check(L.vlm_llm_set_kv(self._handle, C.byref(kv)), "llm_set_kv")
        # row_penalties: the step applies every row's own logits processors (device tables of the state) before sampling
        args = st.args(B=B, with_logprobs=with_logprobs, penalties="rows" if row_penalties else None, **sampler_args)
        if use_graph:
            key = ("rows", B, st.nsplit, block_table.data_ptr(), with_logprobs, bool(row_penalties), bool(q8), tuple(sorted(sampler_args.items())))
            if st.graph_key != key or getattr(self, "_graph_owner", None) is not st:
                check(L.vlm_llm_decode_graph_build(self._handle, C.byref(args), stream), "decode_graph_build")
                st.graph_key = key
                self._graph_owner = st
            check(L.vlm_llm_decode_graph_launch(self._handle, stream), "decode_graph_launch")
        else:
            check(L.vlm_llm_decode_step(self._handle, C.byref(args), stream), "decode_step")

    def decode_forward_rows(self, st: DecodeState, B: int, block_table: torch.Tensor, q8: bool = False):
        """The forward half of decode_step_rows, eager: logits of rows 0..B-1 land in st.logits; nothing is sampled, nothing
        advanced (vlm_llm_decode_forward).  The caller finishes the step with its own processors / sampler and
        decode_advance_rows - the route for Python callables, which cannot live inside a captured step."""
        L = _lib.lib()
        pool = self.pool
        q = (pool.kpool8.data_ptr(), pool.vpool8.data_ptr(), pool.ksb.data_ptr(), pool.vsb.data_ptr()) if q8 else (None,) * 4
        kv = _lib.KvPool(pool.kpool.data_ptr(), pool.vpool.data_ptr(), pool.layer_stride, block_table.data_ptr(),
                         pool.max_pages, *q, 1 if q8 else 0)
        check(L.vlm_llm_set_kv(self._handle, C.byref(kv)), "llm_set_kv")
        args = st.args(B=B)
        check(L.vlm_llm_decode_forward(self._handle, C.byref(args), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "decode_forward")

    def decode_advance_rows(self, st: DecodeState, B: int):
        """ctx += 1, pos += 1, token ring, sampling step counter of rows 0..B-1 (vlm_decode_advance): the tail of an eager step"""
        check(_lib.lib().vlm_decode_advance(st.ctx.data_ptr(), st.pos.data_ptr(), st.tok.data_ptr(), st.out_ring.data_ptr(),
                                            st.ring_len, st.step.data_ptr(), int(B),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "decode_advance")

    # ------------------------------------------------------------------ module contract (reference language.py:404-518)
    def __call__(self, inputs, inputs_embeds=None, mask=None, cache=None, **kwargs):
        position_ids = kwargs.pop("position_ids", None)
        pixel_values = kwargs.pop("pixel_values", None)
        image_grid_thw = kwargs.pop("image_grid_thw", None)
        video_grid_thw = kwargs.pop("video_grid_thw", None)
        rope_deltas_kw = kwargs.pop("rope_deltas", None)
        logits_to_keep = kwargs.pop("logits_to_keep", None)
        if pixel_values is not None:
            self._rope_deltas = None
            self._position_ids = None
        if rope_deltas_kw is not None:
            self._rope_deltas = _to_np(rope_deltas_kw)
        ids = _to_np(inputs)
        if ids.ndim == 1:
            ids = ids[None]
        B, Lq = ids.shape
        if cache is None:
            cache = self.make_cache()
            if B != 1:
                raise NotImplementedError("cache=None is only supported for B == 1")
        caches = [cache] if isinstance(cache[0], KVCache) else cache
        cache_offset = caches[0][0].offset

        # decode widths of the engine: 1 / 2 / 4 / 8 rows on the v_dot2c GEMVs, 9..16 rows on the skinny-M MFMA GEMM, 17..64 as
        # WIDE steps on the prefill GEMMs (these walk the pool's block table in either layout)
        if Lq == 1 and cache_offset > 0 and inputs_embeds is None and (B in (1, 2, 4, 8) or 9 <= B <= self.MAX_DECODE_ROWS):
            # decode: pos = cache offset + rope delta (language.py:476-509); logits only
            deltas = self._rope_deltas if self._rope_deltas is not None else np.zeros((B, 1), dtype=np.int64)
            deltas = np.broadcast_to(np.asarray(deltas).reshape(-1, 1), (B, 1)) if np.asarray(deltas).size == 1 else deltas
            rope_pos = None
            rot = [c[0]._seq for c in caches if c[0]._seq.rotating]
            saved = [(s, s.held, None if s.ring is None else list(s.ring), s.ring_idx) for s in rot]
            try:
                if rot:
                    rope_pos = self._rotate_windows([c[0]._seq for c in caches])
                st = self.decode_begin(caches, ids.reshape(-1), deltas[:B], max_new_tokens=1, rope_pos=rope_pos)
                kv = self._kv_struct(st.seq_row0, decode=B <= 16, q8=self._seqs_q8(st.seqs))
                check(_lib.lib().vlm_llm_set_kv(self._handle, C.byref(kv)), "llm_set_kv")
                args = st.args()
                check(_lib.lib().vlm_llm_decode_forward(self._handle, C.byref(args),
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "decode_forward")
            except Exception:
                # the step did not get enqueued (slot rows, pool exhaustion, ...): the ring goes back to what it was, so a retry
                # plans the SAME move again (newest entry -> the slot of the token that leaves: idempotent on the pool)
                for s, held, ring, ring_idx in saved:
                    s.held, s.ring, s.ring_idx = held, ring, ring_idx
                raise
            for s in st.seqs:
                s.offset += 1
                s.note_decode_step()
            return LanguageModelOutput(logits=st.logits.clone().view(B, 1, -1))

        # prefill path
        if position_ids is None:
            if self._position_ids is not None and cache_offset > 0:
                p = self._position_ids
                position_ids = p[..., cache_offset:cache_offset + Lq]
            else:
                position_ids, rope_deltas = self.get_rope_index(ids, image_grid_thw, video_grid_thw, _to_np(mask))
                self._rope_deltas = rope_deltas
                self._position_ids = position_ids
        pos = _to_np(position_ids)
        if pos.ndim == 2:
            pos = np.broadcast_to(pos[None], (3,) + pos.shape)
        if pos.shape[-1] > Lq:
            pos = pos[..., cache_offset:cache_offset + Lq]
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(ids)
        emb = inputs_embeds.reshape(B * Lq, -1)
        pos_flat = pos.reshape(3, B * Lq)
        logits = self.prefill(emb, pos_flat, caches, [Lq] * B, logits_rows="last" if logits_to_keep == 1 else "all")
        if logits_to_keep == 1:
            return LanguageModelOutput(logits=logits.view(B, 1, -1))
        return LanguageModelOutput(logits=logits.view(B, Lq, -1))
