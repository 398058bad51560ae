"""Phi-3 decoder of Phi-3.5-vision on the decode / prefill engine - host mirror of the language half of the reference's
`mlx_vlm/models/phi3_v/phi3_v.py` (Attention 17-94: one bias-free qkv_proj split [q | k | v], SuScaledRoPE at the cache
offset; MLP 96-106: one gate_up_proj split [gate | up], silu(gate) * up; TransformerBlock 109-133; Phi3V 136-171; lm_head
177) and of `models/rope_utils.py:96-189` (SuScaledRoPE).

The engine (`csrc/engine.hip`) is built for 128-wide heads, rotate-half pairs (d, d + 64), q / k / v rows in one matrix
and interleaved gate / up rows.  Phi-3.5 has 96-wide heads; everything maps onto the engine at load time, bf16 and MLX
4-bit checkpoints alike, without touching a weight's value:

  * q / k rows: real dims [0, 48) -> columns [0, 48), real dims [48, 96) -> columns [64, 112) of the head's 128-wide
    slot (the engine pairs column d with d + 64, i.e. real d with d + 48); zero rows elsewhere.  4-bit: whole packed rows
    move, a zero row is (q = 0, scale = 0, bias = 0).
  * v rows / o_proj columns: head h's 96 dims sit CONTIGUOUSLY at column 128 h + (96 h mod 64) - offset 0 for even heads,
    32 for odd ones.  With that offset every 64-wide quantization group of the 4-bit o_proj (groups run along its input
    dimension, 96 h + d) lands whole inside ONE 64-wide group of the 128-per-head layout, so the packed words and the
    (scale, bias) pairs are moved, never re-quantized; the padding columns multiply attention outputs that are exactly 0.
  * SuScaledRoPE = frequencies 1 / (factor_i * theta ** (2 i / 96)) in the engine's table (48 real pairs, 16 zero) plus
    x * T(scale) on q and k before the rotation - a typed multiply, i.e. one more bf16 rounding of q and k, which attention
    amplifies to ~1 % of an output (measured in tests/test_phi3v_cpu.py), so it is kept: `rope_qk_scale` of
    `vlm_llm_config` applies it with the reference's rounding in the prefill rope pass and in every decode qkv epilogue.
    The reference decides short vs long factors PER CALL: long iff (largest cache offset of the call) + (tokens of the
    call) exceeds `original_max_position_embeddings`, for the rows of that call only - a prompt of 5000 tokens is rotated
    with the long factors, a generation that crosses position 4096 switches at that step while the keys cached before keep
    the short-factor rotation.  Here the engine reads ONE frequency table whose CONTENTS are switched in stream order
    between the calls (the pointer never changes, so captured decode graphs stay valid): `prefill` picks the regime of its
    call, `decode_run` splits a run of steps at the crossing.  Built for one sequence at a time (generate / stream_generate,
    one stream); a continuous batch whose rows would need the long table raises NotImplementedError (its admissions run on a
    second stream and would race with the table).
  * plain RoPE = M-RoPE with equal axes: positions arange(L), rope_deltas 0.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict

import numpy as np
import torch

from .. import quantized as Qz
from ..qwen2_vl.language import LanguageModel as _Engine

ENGINE_HEAD_DIM = 128


def su_scale(max_pos: int, orig_max_pos: int) -> float:
    """SuScaledRoPE.__init__ (rope_utils.py:139-153) default mscale, as the bf16 model sees it (`scale.astype(x.dtype)`)"""
    factor = max_pos / orig_max_pos
    s = 1.0 if factor <= 1.0 else math.sqrt(1 + math.log(factor) / math.log(orig_max_pos))
    return float(torch.tensor(s, dtype=torch.float32).to(torch.bfloat16).to(torch.float32))


class LanguageModel(_Engine):
    ROTATING_POS_FROM_RING = False     # rope offset of a decode step over a rotating cache = cache.offset (the reference's own read)
    ROTATING_PROMPT_WINDOW_MASK = True # phi3_v.py:163 passes cache[0] to create_attention_mask: a prompt > max_kv_size is windowed
    MAX_DECODE_ROWS = 64      # wide steps too: SuScaledRoPE's per-call regime is decided on the device in every rope site
                              # (the fused qkv kernels of the <= 16-row steps, vlm_mrope_kvwrite_decode of the wide ones)

    def __init__(self, config, device="cuda", **engine_kwargs):
        """`config`: the ModelConfig (the reference's Phi3V reads the text parameters from the root)"""
        c = config
        hd = c.hidden_size // c.num_attention_heads
        if hd > ENGINE_HEAD_DIM or hd % 2:
            raise NotImplementedError(f"head_dim {hd}")
        if c.rope_traditional or float(c.partial_rotary_factor) != 1.0:
            raise NotImplementedError("rope_traditional / partial rotary are outside the built path")
        rs = c.rope_scaling or {}
        su = rs.get("type") in ("su", "longrope") or ("short_factor" in rs and "long_factor" in rs)
        if rs and not su:
            raise NotImplementedError(f"rope_scaling {rs.get('type')}: only Su-scaled RoPE (short / long factors) is built")
        base = float(c.rope_theta) ** (np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd))
        freqs, scale = base, 1.0
        self._inv_tables = None
        if su:
            freqs = np.asarray(rs["short_factor"], dtype=np.float32) * base
            scale = su_scale(c.max_position_embeddings, c.original_max_position_embeddings)
            pad = np.zeros(ENGINE_HEAD_DIM // 2 - hd // 2, dtype=np.float32)
            self._inv_tables = (np.concatenate([np.float32(1.0) / freqs, pad]),
                                np.concatenate([np.float32(1.0) / (np.asarray(rs["long_factor"], dtype=np.float32) * base), pad]))
        self.real_head_dim = hd
        self.model_config = c
        self.max_context = int(c.original_max_position_embeddings) if su else None
        # both frequency tables live on the device (short, then long); the decode qkv epilogues pick the long one for a whole
        # step when any row's cache offset has reached original_max, a prefill call is told its regime (_prefill_rope_long)
        self.rope_long_from = self.max_context or 0
        eng = SimpleNamespace(model_type="phi3_v", hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers,
                              intermediate_size=c.intermediate_size, num_attention_heads=c.num_attention_heads,
                              num_key_value_heads=c.num_key_value_heads, rms_norm_eps=c.rms_norm_eps,
                              vocab_size=c.vocab_size, rope_theta=c.rope_theta, rope_scaling=None,
                              tie_word_embeddings=bool(getattr(c, "tie_word_embeddings", False)),
                              head_dim=ENGINE_HEAD_DIM if hd != ENGINE_HEAD_DIM else None, rope_dim=hd,
                              inv_freq=(np.float32(1.0) / freqs).tolist(),
                              inv_freq_long=self._inv_tables[1][: hd // 2].tolist() if su else None, attn_scale=float(hd) ** -0.5,
                              rope_qk_scale=scale if scale != 1.0 else None)
        super().__init__(eng, config, device=device, **engine_kwargs)

    # ------------------------------------------------------------------ head layouts
    def _qk_index(self, heads: int) -> np.ndarray:
        """engine row -> source row of a [heads * hd] projection, -1 = zero row (rotary halves at columns 0 and 64)"""
        hd, half, E = self.real_head_dim, self.real_head_dim // 2, ENGINE_HEAD_DIM
        idx = np.full((heads, E), -1, dtype=np.int64)
        src = np.arange(heads * hd).reshape(heads, hd)
        idx[:, :half] = src[:, :half]
        idx[:, E // 2: E // 2 + half] = src[:, half:]
        return idx.reshape(-1)

    def _v_index(self, heads: int) -> np.ndarray:
        """engine row -> source row for v (and engine column -> source column of o_proj): head h contiguous at
        128 h + (hd h mod 64)"""
        hd, E = self.real_head_dim, ENGINE_HEAD_DIM
        idx = np.full((heads, E), -1, dtype=np.int64)
        for h in range(heads):
            off = (hd * h) % Qz.GROUP if hd != E else 0
            idx[h, off: off + hd] = np.arange(h * hd, (h + 1) * hd)
        return idx.reshape(-1)

    @staticmethod
    def _rows(w, idx: np.ndarray):
        """gather rows with -1 = zero row; tensors and packed 4-bit matrices alike"""
        dev = w.wq.device if isinstance(w, Qz.QuantW) else w.device
        sel = torch.as_tensor(np.where(idx < 0, 0, idx), device=dev)
        zero = torch.as_tensor(idx < 0, device=dev)
        if isinstance(w, Qz.QuantW):
            wq, sb = w.wq[sel].clone(), w.sb[sel].clone()
            wq[zero] = 0
            sb[zero] = 0
            return Qz.QuantW(wq.contiguous(), sb.contiguous())
        out = w[sel].clone()
        out[zero] = 0
        return out

    def _o_cols(self, w, heads: int):
        """o_proj [D, heads * hd] -> [D, heads * 128] in the v layout.  4-bit: 8-weight words and 64-weight groups move
        whole (hd and the head offsets are multiples of 32)."""
        idx = self._v_index(heads)
        if not isinstance(w, Qz.QuantW):
            return self._rows(w.t(), idx).t().contiguous()
        widx = idx.reshape(-1, 8)                                   # engine word -> source word (columns come in runs of >= 32)
        assert ((widx >= 0).all(1) | (widx < 0).all(1)).all() and (widx[:, 0] % 8 == 0)[widx[:, 0] >= 0].all()
        wsel = np.where(widx[:, 0] < 0, 0, widx[:, 0] // 8)
        dev = w.wq.device
        wq = w.wq[:, torch.as_tensor(wsel, device=dev)].clone()
        wq[:, torch.as_tensor(widx[:, 0] < 0, device=dev)] = 0
        gidx = idx.reshape(-1, Qz.GROUP)                            # engine group -> the ONE source group its real columns share
        gsel = np.zeros(gidx.shape[0], dtype=np.int64)
        for j, row in enumerate(gidx):
            src = np.unique(row[row >= 0] // Qz.GROUP)
            if len(src) > 1:
                raise NotImplementedError("o_proj quantization groups straddle heads in the engine layout")
            gsel[j] = src[0] if len(src) else 0
        return Qz.QuantW(wq.contiguous(), w.sb[:, torch.as_tensor(gsel, device=dev)].contiguous())

    # ------------------------------------------------------------------ weights
    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: the language half of the checkpoint under the reference's names (`model.embed_tokens`, `model.layers.N.
        self_attn.{qkv_proj,o_proj}`, `mlp.{gate_up_proj,down_proj}`, the norms, `lm_head`), bf16 tensors or MLX 4-bit
        triples (`.weight` uint32 / `.scales` / `.biases`)."""
        c = self.model_config
        H, Hkv, hd, I = c.num_attention_heads, c.num_key_value_heads, self.real_head_dim, c.intermediate_size

        def lin(path):
            return Qz.take(W, path) if Qz.has_scales(W, path) else W[path + ".weight"]

        def rows(w, a, b):
            return w.rows(slice(a, b)) if isinstance(w, Qz.QuantW) else w[a:b]

        out: Dict[str, object] = {}
        qi, ki, vi = self._qk_index(H), self._qk_index(Hkv), self._v_index(Hkv)
        for i in range(c.num_hidden_layers):
            p = f"model.layers.{i}."
            qkv = lin(p + "self_attn.qkv_proj")
            out[p + "self_attn.q_proj.weight"] = self._rows(rows(qkv, 0, H * hd), qi)
            out[p + "self_attn.k_proj.weight"] = self._rows(rows(qkv, H * hd, (H + Hkv) * hd), ki)
            out[p + "self_attn.v_proj.weight"] = self._rows(rows(qkv, (H + Hkv) * hd, (H + 2 * Hkv) * hd), vi)
            for n, heads in (("q_proj", H), ("k_proj", Hkv), ("v_proj", Hkv)):
                out[p + f"self_attn.{n}.bias"] = torch.zeros(heads * ENGINE_HEAD_DIM, dtype=torch.bfloat16)
            out[p + "self_attn.o_proj.weight"] = self._o_cols(lin(p + "self_attn.o_proj"), H)
            gu = lin(p + "mlp.gate_up_proj")
            out[p + "mlp.gate_proj.weight"] = rows(gu, 0, I)
            out[p + "mlp.up_proj.weight"] = rows(gu, I, 2 * I)
            out[p + "mlp.down_proj.weight"] = lin(p + "mlp.down_proj")
            out[p + "input_layernorm.weight"] = W[p + "input_layernorm.weight"]
            out[p + "post_attention_layernorm.weight"] = W[p + "post_attention_layernorm.weight"]
        out["model.embed_tokens.weight"] = lin("model.embed_tokens")
        out["model.norm.weight"] = W["model.norm.weight"]
        if not getattr(c, "tie_word_embeddings", False):
            out["lm_head.weight"] = lin("lm_head")
        return super().load_weights(out)

    # ------------------------------------------------------------------ positions: plain RoPE
    def get_rope_index(self, input_ids, image_grid_thw=None, video_grid_thw=None, attention_mask=None):
        ids = np.asarray(input_ids)
        B, L = ids.shape
        pos = np.broadcast_to(np.arange(L, dtype=np.int64)[None, None], (3, B, L)).copy()
        return pos, np.zeros((B, 1), dtype=np.int64)

    # ------------------------------------------------------------------ short / long factor regimes (rope_utils.py:168-172)
    # SuScaledRoPE decides PER CALL: position_end = max(cache offset over the rows of the call) + tokens of the call;
    # long factors (for every row of the call) iff position_end > original_max_position_embeddings.
    #   * decode steps: evaluated inside the qkv epilogue from the rows' cache offsets (vlm_llm_config.rope_long_from) -
    #     a captured step crosses the limit by itself, a continuous batch switches all its rows when its longest row
    #     crosses, exactly as the reference's batched call does;
    #   * prefill calls (single prompt, chunk onto a cache, an admission of several prompts): the regime of the call is
    #     computed here and handed to the engine (vlm_prefill_args.rope_long).
    def _call_regime(self, caches, lengths) -> bool:
        if self.max_context is None:
            return False
        offs = [int(cch[0]._seq.offset) for cch in caches]
        return max(offs) + max(int(n) for n in lengths) > self.max_context

    def prefill(self, inputs_embeds, position_ids, caches, lengths, logits_rows="last", reserve_extra=0):
        self._prefill_rope_long = self._call_regime(caches, lengths)
        try:
            return super().prefill(inputs_embeds, position_ids, caches, lengths, logits_rows, reserve_extra)
        finally:
            self._prefill_rope_long = False
