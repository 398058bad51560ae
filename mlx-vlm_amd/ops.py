"""Thin torch-tensor front end of the C ABI (include/vlm_hip.h).

torch is plumbing only here: it owns device memory and the HIP stream; every
computation below is a call into libvlm_hip.so.  Tensors must be CUDA(HIP)
resident, contiguous along the last dim and bf16 unless stated otherwise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check

EPI_NONE, EPI_BIAS, EPI_GELU_FAST, EPI_GELU_ERF, EPI_RESIDUAL, EPI_SWIGLU = 0, 1, 2, 4, 8, 16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.VlmHipError("libvlm_hip ops need device tensors (there is no CPU fallback)")


def gemm(a, w, bias=None, res=None, out=None, epilogue=EPI_NONE):
    """out[M,N(/2)] = epi(a[M,K] @ w[N,K].T)"""
    _dev(a, w, bias, res, out)
    M, K = a.shape
    N = w.shape[0]
    n_out = N // 2 if epilogue & EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=a.device)
    check(_lib.lib().vlm_gemm_bf16(_p(a), _p(w), _p(bias), _p(res), _p(out), M, N, K, a.stride(0), w.stride(0),
                                   out.stride(0), res.stride(0) if res is not None else 0, epilogue, _stream()), "gemm")
    return out


def gemm_rope2d(a, w, bias, cos_sin, head_dim, rope_cols, out=None):
    """out = rope2d(a @ w.T + bias) on the first rope_cols columns (q / k heads with interleaved pairs), bias only on the
    rest (vlm_gemm_bf16_rope2d).  cos_sin fp32 [2, M, head_dim / 2]."""
    _dev(a, w, bias, cos_sin, out)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    check(_lib.lib().vlm_gemm_bf16_rope2d(_p(a), _p(w), _p(bias), _p(cos_sin), _p(out), M, N, K, a.stride(0), w.stride(0),
                                          out.stride(0), int(head_dim), int(rope_cols), _stream()), "gemm_rope2d")
    return out


def gemv(x, w, bias=None, res=None, norm_w=None, out=None, eps=1e-6, epilogue=EPI_NONE):
    _dev(x, w, bias, res, norm_w, out)
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if epilogue & EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().vlm_gemv_bf16(_p(x), _p(w), _p(bias), _p(res), _p(norm_w), _p(out), M, N, K, x.stride(0),
                                   w.stride(0), out.stride(0), res.stride(0) if res is not None else 0, eps, epilogue,
                                   _stream()), "gemv")
    return out


_gemv_ws = {}


def gemv_workspace(device):
    """zero-initialised split-K workspace of the batched (5..16 row) decode projections, one per device for callers of the
    operator layer (the engine owns its own)"""
    key = torch.device(device).index or 0
    if key not in _gemv_ws:
        _gemv_ws[key] = torch.zeros(_lib.lib().vlm_gemv_workspace_bytes(), dtype=torch.uint8, device=device)
    return _gemv_ws[key]


EPI_X_TILED, EPI_Y_TILED = 64, 128      # include/vlm_hip.h: activations in the MFMA tile's layout [K / 8][16][8] between two projections


def tile_rows(x):
    """[M <= 16, K] row-major -> the tiled layout [K / 8, 16, 8] (rows M..15 zero): what EPI_X_TILED reads / EPI_Y_TILED writes"""
    M, K = x.shape
    t = torch.zeros(K // 8, 16, 8, dtype=x.dtype, device=x.device)
    t[:, :M] = x.view(M, K // 8, 8).permute(1, 0, 2)
    return t


def untile_rows(t, M):
    """inverse of tile_rows: [K / 8, 16, 8] -> [M, K]"""
    return t[:, :M].permute(1, 0, 2).reshape(M, -1).contiguous()


def gemv_ws(x, w, bias=None, res=None, norm_w=None, out=None, eps=1e-6, epilogue=EPI_NONE, M=None):
    """gemv for a batched decode step (up to 16 rows), K split over workgroups through the workspace where it pays.
    epilogue | EPI_Y_TILED: out is the tiled [N_out / 8, 16, 8] tensor; epilogue | EPI_X_TILED: x is one (pass M = rows)."""
    _dev(x, w, bias, res, norm_w, out)
    N = w.shape[0]
    n_out = N // 2 if epilogue & EPI_SWIGLU else N
    if epilogue & EPI_X_TILED:
        K, ldx = x.shape[0] * 8, 0
        assert M is not None and x.shape[1:] == (16, 8)
    else:
        M, K = x.shape
        ldx = x.stride(0)
    if out is None:
        out = (torch.full((n_out // 8, 16, 8), float("nan"), dtype=torch.bfloat16, device=x.device) if epilogue & EPI_Y_TILED
               else torch.empty(M, n_out, dtype=torch.bfloat16, device=x.device))
    ldy = 0 if epilogue & EPI_Y_TILED else out.stride(0)
    check(_lib.lib().vlm_gemv_bf16_ws(_p(x), _p(w), _p(bias), _p(res), _p(norm_w), _p(out), M, N, K, ldx,
                                      w.stride(0), ldy, res.stride(0) if res is not None else 0, eps, epilogue,
                                      _p(gemv_workspace(x.device)), _stream()), "gemv_ws")
    return out


def gemv_qkv_rope_kvwrite_ws(h, norm_w, wqkv, bqkv, Hq, Hkv, D, pos, slot, inv_freq, block_table, kpool, vpool, eps=1e-6,
                             out=None, max_pages=None):
    _dev(h, norm_w, wqkv, bqkv, pos, slot, inv_freq, block_table, kpool, vpool)
    M, K = h.shape
    if out is None:
        out = torch.zeros(M, (Hq + 2 * Hkv) * D, dtype=torch.bfloat16, device=h.device)
    check(_lib.lib().vlm_gemv_qkv_rope_kvwrite_ws(_p(h), _p(norm_w), eps, _p(wqkv), _p(bqkv), _p(out), out.stride(0), M, K,
                                                  Hq, Hkv, D, _p(pos), _p(slot), _p(inv_freq), _p(block_table),
                                                  block_table.shape[1] if block_table is not None else int(max_pages),
                                                  _p(kpool), _p(vpool), _p(gemv_workspace(h.device)), _stream()),
          "gemv_qkv_rope_kvwrite_ws")
    return out


def gemv_qkv_rope_kvwrite(h, norm_w, wqkv, bqkv, Hq, Hkv, D, pos, slot, inv_freq, block_table, kpool, vpool, eps=1e-6,
                          out=None, max_pages=None):
    """decode-step fusion: RMSNorm + qkv GEMV + bias + M-RoPE + paged KV write.  -> qkv [M, (Hq+2Hkv)*D] (q part valid)
    block_table None: identity layout (row m owns pages [m*max_pages, (m+1)*max_pages) of the pools as passed)."""
    _dev(h, norm_w, wqkv, bqkv, pos, slot, inv_freq, block_table, kpool, vpool)
    M, K = h.shape
    if out is None:
        out = torch.zeros(M, (Hq + 2 * Hkv) * D, dtype=torch.bfloat16, device=h.device)
    check(_lib.lib().vlm_gemv_qkv_rope_kvwrite(_p(h), _p(norm_w), eps, _p(wqkv), _p(bqkv), _p(out), out.stride(0), M, K,
                                               Hq, Hkv, D, _p(pos), _p(slot), _p(inv_freq), _p(block_table),
                                               block_table.shape[1] if block_table is not None else int(max_pages),
                                               _p(kpool), _p(vpool), _stream()), "gemv_qkv_rope_kvwrite")
    return out


def gemv_attn_out_(part_o, part_ml, wo, h, Hq, D):
    """h += merge(partials) @ wo.T   (in place on the residual stream)"""
    _dev(part_o, part_ml, wo, h)
    M, N = h.shape
    check(_lib.lib().vlm_gemv_attn_out(_p(part_o), _p(part_ml), part_o.shape[2], _p(wo), _p(h), h.stride(0), M, N, Hq, D,
                                       _stream()), "gemv_attn_out")
    return h


def layernorm(x, w, b, eps=1e-6, out=None):
    _dev(x, w, b)
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().vlm_layernorm(_p(x), _p(w), _p(b), _p(out), x.shape[0], x.shape[1], eps, _stream()), "layernorm")
    return out


def rmsnorm(x, w, eps=1e-6, res=None, out=None, h_out=None):
    _dev(x, w, res)
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().vlm_rmsnorm_residual(_p(x), _p(res), _p(w), _p(out), _p(h_out), x.shape[0], x.shape[1], eps,
                                          _stream()), "rmsnorm")
    return out


def rope2d_vision_(qkv, cos_tab, sin_tab, n_heads):
    """in place on qkv [N, 3*H*D]; cos/sin fp32 [N, D/2]"""
    _dev(qkv, cos_tab, sin_tab)
    N = qkv.shape[0]
    D = qkv.shape[1] // (3 * n_heads)
    check(_lib.lib().vlm_rope2d_vision(_p(qkv), _p(cos_tab), _p(sin_tab), N, n_heads, D, qkv.stride(0), _stream()), "rope2d")
    return qkv


def mrope_kvwrite_(qkv, Hq, Hkv, D, pos_t, pos_h, pos_w, inv_freq, sec0, sec1, kv_seq=None, kv_slot=None,
                   block_table=None, kpool=None, vpool=None, qk_scale=None):
    """qk_scale: SuScaledRoPE's typed input scale of q and k (vlm_mrope_kvwrite_scaled); None = plain"""
    _dev(qkv, pos_t, inv_freq)
    T = qkv.shape[0]
    max_pages = block_table.shape[1] if block_table is not None else 0
    if qk_scale is not None and float(qk_scale) != 1.0:
        check(_lib.lib().vlm_mrope_kvwrite_scaled(_p(qkv), qkv.stride(0), T, Hq, Hkv, D, _p(pos_t), _p(pos_h), _p(pos_w),
                                                  _p(inv_freq), sec0, sec1, _p(kv_seq), _p(kv_slot), _p(block_table),
                                                  max_pages, _p(kpool), _p(vpool), float(qk_scale), _stream()),
              "mrope_kvwrite_scaled")
        return qkv
    check(_lib.lib().vlm_mrope_kvwrite(_p(qkv), qkv.stride(0), T, Hq, Hkv, D, _p(pos_t), _p(pos_h), _p(pos_w),
                                       _p(inv_freq), sec0, sec1, _p(kv_seq), _p(kv_slot), _p(block_table), max_pages,
                                       _p(kpool), _p(vpool), _stream()), "mrope_kvwrite")
    return qkv


def kv_gather_(qkv, Hq, Hkv, D, kv_slot, block_table, kpool, vpool, kv_seq=None):
    """cached k / v of (kv_seq[t], kv_slot[t]) -> the k / v columns of row t of qkv [T, (Hq + 2 Hkv) * D] (in place)"""
    _dev(qkv, kv_slot, block_table, kpool, vpool, kv_seq)
    check(_lib.lib().vlm_kv_gather(_p(qkv), qkv.stride(0), qkv.shape[0], Hq, Hkv, D, _p(kv_seq), _p(kv_slot), _p(block_table),
                                   block_table.shape[1], _p(kpool), _p(vpool), _stream()), "kv_gather")
    return qkv


def attn_prefill(q, k, v, cu_seqlens, total_qblocks, Hq, Hkv, D, scale, causal, out=None, uniform_segments=False, q_start=None):
    """q/k/v: 2-D views [T, *] whose data_ptr points at head 0 of token 0 and stride(0) is the token stride.
    uniform_segments: placement hint (all segments the same length); results are identical.
    q_start (int32 [nseg], device): rows of segment s before q_start[s] are keys only (a cached prefix) - their rows of `out`
    are not written; total_qblocks then counts sum ceil((len_s - q_start_s) / 128)."""
    _dev(q, k, v, cu_seqlens, q_start)
    T = q.shape[0]
    if out is None:
        out = torch.empty(T, Hq * D, dtype=torch.bfloat16, device=q.device)
    nseg = cu_seqlens.numel() - 1
    flags = (1 if causal else 0) | (2 if uniform_segments else 0)
    if q_start is not None:
        cu_seqlens = torch.cat([cu_seqlens.reshape(-1).to(torch.int32), q_start.reshape(-1).to(torch.int32)])     # [cu | q_start]
        flags |= 4
    check(_lib.lib().vlm_attn_prefill(_p(q), _p(k), _p(v), _p(out), q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                      _p(cu_seqlens), nseg, total_qblocks, Hq, Hkv, D, scale, flags, _stream()), "attn_prefill")
    return out


def attn_decode_paged(q, kpool, vpool, block_table, kv_len, kv_len_add, Hq, Hkv, D, scale, nsplit, out=None,
                      merge=True, max_pages=None):
    """merge=True -> bf16 [B, Hq*D]; merge=False -> (part_o, part_ml) for gemv_attn_out_
    block_table None: identity layout (row b owns pages [b*max_pages, (b+1)*max_pages) of the pools as passed)."""
    _dev(q, kpool, vpool, block_table, kv_len)
    B = q.shape[0]
    part_o = torch.empty(B, Hq, nsplit, D, dtype=torch.float32, device=q.device)
    part_ml = torch.empty(B, Hq, nsplit, 2, dtype=torch.float32, device=q.device)
    if merge and out is None:
        out = torch.empty(B, Hq * D, dtype=torch.bfloat16, device=q.device)
    check(_lib.lib().vlm_attn_decode_paged(_p(q), q.stride(0), _p(kpool), _p(vpool), _p(block_table),
                                           block_table.shape[1] if block_table is not None else int(max_pages),
                                           _p(kv_len), kv_len_add, B, Hq, Hkv, D, scale, nsplit,
                                           _p(part_o), _p(part_ml), _p(out) if merge else None,
                                           out.stride(0) if merge else 0, _stream()), "attn_decode")
    return out if merge else (part_o, part_ml)


def gemv_attn_out_bf16_(part_o, part_ml, wo, h, Hq, D):
    """h[0] += merge(page-split partials) @ wo.T in place (one decode row)"""
    _dev(part_o, part_ml, wo, h)
    check(_lib.lib().vlm_gemv_attn_out_bf16(_p(part_o), _p(part_ml), part_o.shape[2], _p(wo), _p(h), h.stride(0), wo.shape[0],
                                            Hq, D, _stream()), "gemv_attn_out_bf16")
    return h


def attn_decode_paged_split(q, kpool, vpool, block_table, kv_len, kv_len_add, Hq, Hkv, D, scale, nsplit, out=None,
                            max_pages=None, tickets=None, merge=True, part=None):
    """page-split decode attention (one wave per page stride).  merge=True: the last arriver merges -> bf16 [B, Hq*D];
    merge=False: the partial-only form -> (part_o fp32 [B, Hq, nsplit, D], part_ml fp32 [B, Hq, nsplit, 2]) for
    gemv_attn_out_bf16_."""
    _dev(q, kpool, vpool, block_table, kv_len)
    B = q.shape[0]
    if not merge:
        if part is not None:
            part_o, part_ml = part
        else:
            part_o = torch.full((B, Hq, nsplit, D), float("nan"), dtype=torch.float32, device=q.device)   # unwritten = NaN on purpose
            part_ml = torch.empty(B, Hq, nsplit, 2, dtype=torch.float32, device=q.device)
        check(_lib.lib().vlm_attn_decode_paged_split(_p(q), q.stride(0), _p(kpool), _p(vpool), _p(block_table),
                                                     block_table.shape[1] if block_table is not None else int(max_pages),
                                                     _p(kv_len), kv_len_add, B, Hq, Hkv, D, scale, nsplit, _p(part_o),
                                                     _p(part_ml), None, None, 0, _stream()), "attn_decode_split")
        return part_o, part_ml
    part_o = torch.empty(B, Hq, nsplit, D, dtype=torch.float32, device=q.device)
    part_ml = torch.empty(B, Hq, nsplit, 2, dtype=torch.float32, device=q.device)
    if tickets is None:
        tickets = torch.zeros(B * Hkv, dtype=torch.int32, device=q.device)
    if out is None:
        out = torch.empty(B, Hq * D, dtype=torch.bfloat16, device=q.device)
    check(_lib.lib().vlm_attn_decode_paged_split(_p(q), q.stride(0), _p(kpool), _p(vpool), _p(block_table),
                                                 block_table.shape[1] if block_table is not None else int(max_pages),
                                                 _p(kv_len), kv_len_add, B, Hq, Hkv, D, scale, nsplit, _p(part_o),
                                                 _p(part_ml), _p(tickets), _p(out), out.stride(0), _stream()),
          "attn_decode_split")
    return out


class EncoderLayers:
    """The weights of a SigLIP / CLIP encoder stack as the native layer loop takes them (vlm_enc_layer array, built once at
    load from a tower's `_w` dict: keys "<i>.ln1w", "<i>.ln1b", "<i>.wqkv", "<i>.bqkv", "<i>.wo", "<i>.bo", "<i>.ln2w",
    "<i>.ln2b", "<i>.w1", "<i>.b1", "<i>.w2", "<i>.b2") + the per-call workspaces, cached by token count."""

    def __init__(self, w: dict, n_layers: int):
        self.n = n_layers
        self._keep = [w[f"{i}.{k}"].contiguous() for i in range(n_layers) for k in ("ln1w", "ln1b", "wqkv", "bqkv", "wo", "bo", "ln2w",
                                                                                    "ln2b", "w1", "b1", "w2", "b2")]
        for t in self._keep:
            if not (t.is_cuda and t.dtype == torch.bfloat16):
                raise ValueError("encoder weights must be bf16 device tensors")
        self.arr = (_lib.EncLayer * n_layers)(*[_lib.EncLayer(*[t.data_ptr() for t in self._keep[12 * i: 12 * i + 12]])
                                                for i in range(n_layers)])
        self._ws = {}

    def forward_(self, x, H, head_dim, ln_eps, act_epilogue, cu_seqlens, total_qblocks, scale, uniform_segments=True):
        """x [N, E] bf16 (device, contiguous) is the residual stream: overwritten with the output of the last layer"""
        _dev(x, cu_seqlens)
        N, E = x.shape
        MH = self._keep[8].shape[0]                 # fc1 rows
        key = (N, x.device, torch.cuda.current_stream().cuda_stream)      # (an admission prefill may run on a side stream)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) > 4:
                self._ws.clear()
            mk = lambda c: torch.empty(N, c, dtype=torch.bfloat16, device=x.device)   # noqa: E731
            ws = self._ws[key] = (mk(E), mk(3 * H * head_dim), mk(H * head_dim), mk(MH))
        check(_lib.lib().vlm_encoder_forward(self.arr, self.n, _p(x), _p(ws[0]), _p(ws[1]), _p(ws[2]), _p(ws[3]), N, E, H, head_dim,
                                             MH, float(ln_eps), int(act_epilogue), _p(cu_seqlens), cu_seqlens.numel() - 1,
                                             int(total_qblocks), float(scale), int(bool(uniform_segments)), _stream()),
              "encoder_forward")
        return x


def gemm_w4(a, wq, sb, bias=None, res=None, epilogue=EPI_NONE, out=None):
    """prefill GEMM over MLX 4-bit weights with the dequantisation fused into the tile staging: wq int32 [N, K/8], sb int32
    [N, K/64] (as gemv_w4)"""
    _dev(a, wq, sb, bias, res, out)
    M, K = a.shape
    N = wq.shape[0]
    n_out = N // 2 if epilogue & EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=a.device)
    check(_lib.lib().vlm_gemm_w4(_p(a), _p(wq), _p(sb), _p(bias), _p(res), _p(out), M, N, K, a.stride(0), out.stride(0),
                                 res.stride(0) if res is not None else 0, epilogue, _stream()), "gemm_w4")
    return out


def kv_quantize_tokens(kpool, vpool, kpool8, vpool8, ksb, vsb, kv_seq, kv_slot, block_table, Hkv, D, n_layers=1,
                       layer_stride=0, max_pages=None):
    """KVCache.to_quantized over listed tokens: bf16 pools -> 8-bit pools (u8 + (scale | bias) words), all layers"""
    _dev(kpool, vpool, kpool8, vpool8, ksb, vsb, kv_seq, kv_slot, block_table)
    check(_lib.lib().vlm_kv_quantize_tokens(_p(kpool), _p(vpool), _p(kpool8), _p(vpool8), _p(ksb), _p(vsb), int(layer_stride),
                                            int(n_layers), _p(kv_seq), _p(kv_slot), int(kv_slot.numel()), _p(block_table),
                                            block_table.shape[1] if block_table is not None else int(max_pages), Hkv, D,
                                            _stream()), "kv_quantize_tokens")


def attn_decode_paged_q8(q, kpool16, vpool16, kpool8, vpool8, ksb, vsb, block_table, kv_len, kv_len_add, Hq, Hkv, D, scale, nsplit,
                         quantize_new=True, out=None, max_pages=None, tickets=None, merge=True):
    """page-split decode attention over the 8-bit KV pools (quantized_scaled_dot_product_attention at L == 1); merge as
    attn_decode_paged_split"""
    _dev(q, kpool16, vpool16, kpool8, vpool8, ksb, vsb, block_table, kv_len)
    B = q.shape[0]
    mp = block_table.shape[1] if block_table is not None else int(max_pages)
    if not merge:
        part_o = torch.full((B, Hq, nsplit, D), float("nan"), dtype=torch.bfloat16, device=q.device)
        part_ml = torch.empty(B, Hq, nsplit, 2, dtype=torch.float32, device=q.device)
        check(_lib.lib().vlm_attn_decode_paged_q8(_p(q), q.stride(0), _p(kpool16), _p(vpool16), _p(kpool8), _p(vpool8), _p(ksb),
                                                  _p(vsb), _p(block_table), mp, _p(kv_len), kv_len_add, B, Hq, Hkv, D, scale,
                                                  nsplit, _p(part_o), _p(part_ml), None, None, 0, int(bool(quantize_new)),
                                                  _stream()), "attn_decode_q8")
        return part_o, part_ml
    # (2 * nsplit: from 128 (row, kv head) pairs the launch runs half-page units, i.e. twice the splits)
    part_o = torch.empty(B, Hq, 2 * nsplit, D, dtype=torch.float32, device=q.device)
    part_ml = torch.empty(B, Hq, 2 * nsplit, 2, dtype=torch.float32, device=q.device)
    if tickets is None:
        tickets = torch.zeros(B * Hkv, dtype=torch.int32, device=q.device)
    if out is None:
        out = torch.empty(B, Hq * D, dtype=torch.bfloat16, device=q.device)
    check(_lib.lib().vlm_attn_decode_paged_q8(_p(q), q.stride(0), _p(kpool16), _p(vpool16), _p(kpool8), _p(vpool8), _p(ksb),
                                              _p(vsb), _p(block_table), mp, _p(kv_len), kv_len_add, B, Hq, Hkv, D, scale, nsplit,
                                              _p(part_o), _p(part_ml), _p(tickets), _p(out), out.stride(0),
                                              int(bool(quantize_new)), _stream()), "attn_decode_q8")
    return out


def embed_gather(ids, table, out=None):
    _dev(ids, table)
    T = ids.numel()
    if out is None:
        out = torch.empty(T, table.shape[1], dtype=table.dtype, device=table.device)
    check(_lib.lib().vlm_embed_gather(_p(ids), _p(table), _p(out), T, table.shape[1], out.stride(0), table.shape[0],
                                      _stream()), "embed_gather")
    return out


def scatter_rows_(src, dst_rows, dst):
    _dev(src, dst_rows, dst)
    check(_lib.lib().vlm_scatter_image_rows(_p(src), _p(dst_rows), _p(dst), src.shape[0], src.shape[1], src.stride(0),
                                            dst.stride(0), _stream()), "scatter_rows")
    return dst


def cast_pad(src_f32, ld_dst):
    _dev(src_f32)
    rows, cols = src_f32.shape
    out = torch.empty(rows, ld_dst, dtype=torch.bfloat16, device=src_f32.device)
    check(_lib.lib().vlm_cast_f32_bf16_pad(_p(src_f32), _p(out), rows, cols, src_f32.stride(0), ld_dst, _stream()), "cast_pad")
    return out


def sample_workspace(B, device):
    # zeroed: the first 256 bytes hold the arrival ticket of the fused greedy tail (csrc/sample.hip)
    return torch.zeros(_lib.lib().vlm_sample_workspace_bytes(B), dtype=torch.uint8, device=device)


def bad_argmax_rows(ws) -> int:
    """rows of fused greedy tails run over this workspace whose logits held no finite candidate (all NaN): csrc/sample.hip
    counts them behind the ticket and hands out token 0 instead of turning the sentinel into an address"""
    return int(ws[4:8].view(torch.int32).item())


def sample(logits, temperature=0.0, top_p=1.0, min_p=0.0, top_k=0, seed=0, step=None, want_logprobs=True, ws=None,
           min_tokens_to_keep=1, top_n_sigma=0.0, p_less=False, typical_p=1.0, xtc_probability=0.0, xtc_threshold=0.0,
           xtc_special_tokens=None, return_filtered=False, input_is_logprobs=False):
    """-> (tokens int32 [B], logprobs bf16 [B,V] or None[, filtered logprobs bf16 [B,V] when return_filtered]).
    vlm_sample_ex: the whole make_sampler surface (sample_utils.py:10-89), scalars handed over as python floats (doubles).
    input_is_logprobs (temperature > 0): `logits` already holds log-probs - the contract of the reference's sampler closures -
    and is filtered as it is."""
    _dev(logits)
    B, V = logits.shape
    tok = torch.empty(B, dtype=torch.int32, device=logits.device)
    lp_given = bool(input_is_logprobs) and temperature > 0
    lp = torch.empty(B, V, dtype=torch.bfloat16, device=logits.device) if ((want_logprobs or temperature > 0) and not lp_given) else None
    scratch = torch.empty(B, V, dtype=torch.bfloat16, device=logits.device) if temperature > 0 else None
    if ws is None:
        ws = sample_workspace(B, logits.device)
    sp = _lib.SamplerParams(input_is_logprobs=int(lp_given), temperature=float(temperature), top_p=float(top_p), min_p=float(min_p),
                            min_tokens_to_keep=int(min_tokens_to_keep), top_k=int(top_k), top_n_sigma=float(top_n_sigma),
                            p_less=int(bool(p_less)), typical_p=float(typical_p), xtc_probability=float(xtc_probability),
                            xtc_threshold=float(xtc_threshold), seed=int(seed) & 0xFFFFFFFF)
    keep = []
    if temperature > 0 and xtc_probability > 0 and xtc_special_tokens is not None and len(xtc_special_tokens):
        sp_tok = torch.as_tensor([int(t) for t in xtc_special_tokens], dtype=torch.int32).to(logits.device)
        keep.append(sp_tok)
        sp.xtc_special_tokens, sp.n_xtc_special = _p(sp_tok), sp_tok.numel()
    if temperature > 0 and 0.0 < typical_p < 1.0:
        sort_ws = torch.empty(_lib.lib().vlm_sample_sort_workspace_bytes(B, V), dtype=torch.uint8, device=logits.device)
        keep.append(sort_ws)
        sp.sort_workspace = _p(sort_ws)
    check(_lib.lib().vlm_sample_ex(_p(logits), logits.stride(0), B, V, _p(lp), _p(scratch), V, _p(tok), _p(ws), sp, _p(step),
                                   _stream()), "sample")
    del keep                           # (allocated and freed on the launch's own stream: the caching allocator orders reuse)
    if lp_given:
        lp = logits
    if return_filtered:
        # (no filter on: the kernel draws from the log-probs themselves and leaves the scratch row alone)
        any_filter = temperature > 0 and (0.0 < top_p < 1.0 or min_p != 0.0 or top_k > 0 or top_n_sigma > 0.0 or p_less
                                          or 0.0 < typical_p < 1.0 or xtc_probability > 0.0)
        return tok, lp, (scratch if any_filter else lp)
    return tok, lp


def sample_greedy_advance(logits, tok, ctx, pos, step, embed, h, out_ring=None, want_logprobs=True, ws=None):
    """Greedy tail of a decode step (vlm_sample_greedy_advance): tok <- argmax, ctx += 1, pos += 1, ring, step += 1,
    h <- embed[tok].  -> logprobs bf16 [B, V] or None"""
    _dev(logits, tok, ctx, pos, step, embed, h, out_ring)
    B, V = logits.shape
    lp = torch.empty(B, V, dtype=torch.bfloat16, device=logits.device) if want_logprobs else None
    if ws is None:
        ws = sample_workspace(B, logits.device)
    check(_lib.lib().vlm_sample_greedy_advance(_p(logits), logits.stride(0), B, V, _p(lp), V, _p(tok), _p(ws), _p(ctx), _p(pos),
                                               _p(out_ring), out_ring.shape[0] if out_ring is not None else 0, _p(step),
                                               _p(embed), _p(h), embed.shape[1], h.stride(0), _stream()),
          "sample_greedy_advance")
    return lp


def apply_logit_penalties(logits, pen_args, push_tok=None):
    """In place on logits [B, V] (bf16): logit_bias / repetition / presence / frequency penalties over the device token
    history (vlm_apply_logit_penalties).  pen_args: _lib.PenaltyArgs; push_tok int32 [B] is appended to the history first."""
    _dev(logits, push_tok)
    B, V = logits.shape
    check(_lib.lib().vlm_apply_logit_penalties(_p(logits), logits.stride(0), B, V, _p(push_tok), C.byref(pen_args), _stream()),
          "apply_logit_penalties")
    return logits


def gemv_w4(x, wq, sb, bias=None, res=None, norm_w=None, out=None, eps=1e-6, epilogue=EPI_NONE):
    """decode GEMV over MLX 4-bit weights: wq int32 [N, K/8] (uint32 words), sb int32 [N, K/64] (scale | bias << 16)"""
    _dev(x, wq, sb, bias, res, norm_w, out)
    M, K = x.shape
    N = wq.shape[0]
    n_out = N // 2 if epilogue & EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().vlm_gemv_w4(_p(x), _p(wq), _p(sb), _p(bias), _p(res), _p(norm_w), _p(out), M, N, K, x.stride(0),
                                 out.stride(0), res.stride(0) if res is not None else 0, eps, epilogue, _stream()), "gemv_w4")
    return out


def gemv_w4_qkv_rope_kvwrite(h, norm_w, wq, sb, bqkv, Hq, Hkv, D, pos, slot, inv_freq, block_table, kpool, vpool, eps=1e-6,
                             out=None, max_pages=None):
    """gemv_qkv_rope_kvwrite over 4-bit q/k/v rows (wq / sb as gemv_w4)"""
    _dev(h, norm_w, wq, sb, bqkv, pos, slot, inv_freq, block_table, kpool, vpool)
    M, K = h.shape
    if out is None:
        out = torch.zeros(M, (Hq + 2 * Hkv) * D, dtype=torch.bfloat16, device=h.device)
    check(_lib.lib().vlm_gemv_w4_qkv_rope_kvwrite(_p(h), _p(norm_w), eps, _p(wq), _p(sb), _p(bqkv), _p(out), out.stride(0), M,
                                                  K, Hq, Hkv, D, _p(pos), _p(slot), _p(inv_freq), _p(block_table),
                                                  block_table.shape[1] if block_table is not None else int(max_pages),
                                                  _p(kpool), _p(vpool), _stream()), "gemv_w4_qkv_rope_kvwrite")
    return out


def gemv_w4_ws(x, wq, sb, bias=None, res=None, norm_w=None, out=None, eps=1e-6, epilogue=EPI_NONE):
    """gemv_w4 for a batched decode step (up to 16 rows), split-K through the workspace where it pays"""
    _dev(x, wq, sb, bias, res, norm_w, out)
    M, K = x.shape
    N = wq.shape[0]
    n_out = N // 2 if epilogue & EPI_SWIGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().vlm_gemv_w4_ws(_p(x), _p(wq), _p(sb), _p(bias), _p(res), _p(norm_w), _p(out), M, N, K, x.stride(0),
                                    out.stride(0), res.stride(0) if res is not None else 0, eps, epilogue,
                                    _p(gemv_workspace(x.device)), _stream()), "gemv_w4_ws")
    return out


def gemv_w4_qkv_rope_kvwrite_ws(h, norm_w, wq, sb, bqkv, Hq, Hkv, D, pos, slot, inv_freq, block_table, kpool, vpool, eps=1e-6,
                                out=None, max_pages=None):
    _dev(h, norm_w, wq, sb, bqkv, pos, slot, inv_freq, block_table, kpool, vpool)
    M, K = h.shape
    if out is None:
        out = torch.zeros(M, (Hq + 2 * Hkv) * D, dtype=torch.bfloat16, device=h.device)
    check(_lib.lib().vlm_gemv_w4_qkv_rope_kvwrite_ws(_p(h), _p(norm_w), eps, _p(wq), _p(sb), _p(bqkv), _p(out), out.stride(0), M,
                                                     K, Hq, Hkv, D, _p(pos), _p(slot), _p(inv_freq), _p(block_table),
                                                     block_table.shape[1] if block_table is not None else int(max_pages),
                                                     _p(kpool), _p(vpool), _p(gemv_workspace(h.device)), _stream()),
          "gemv_w4_qkv_rope_kvwrite_ws")
    return out


def dequant_w4(wq, sb, rows=None, out=None):
    """bf16 rows of a 4-bit matrix: all of them (rows None) or a gather (embedding lookup)"""
    _dev(wq, sb, rows, out)
    N, K = wq.shape[0], wq.shape[1] * 8
    n = N if rows is None else rows.numel()
    if out is None:
        out = torch.empty(n, K, dtype=torch.bfloat16, device=wq.device)
    check(_lib.lib().vlm_dequant_w4(_p(wq), _p(sb), _p(rows), _p(out), n, K, out.stride(0), N, _stream()), "dequant_w4")
    return out


def gemm_set_staging(mode: int):
    """0 = automatic (LDS DMA when K % 64 == 0), 1 = always register staging (test / A-B knob)."""
    check(_lib.lib().vlm_gemm_set_staging(int(mode)), "gemm_set_staging")
