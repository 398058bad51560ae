"""ctypes binding of libvlm_hip.so (include/vlm_hip.h).

There is NO fallback: if the HIP library is missing or does not export a symbol
the import of any compute path raises - a product path that silently ran on a
CPU/eager substitute would void every parity claim.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLM_HIP_LIB") or os.path.join(_HERE, "lib", "libvlm_hip.so")

c_void_p, c_int, c_float, c_uint, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_uint, C.c_size_t


class LlmConfig(C.Structure):
    _fields_ = [("hidden", c_int), ("n_layers", c_int), ("inter", c_int), ("n_heads", c_int), ("n_kv_heads", c_int),
                ("head_dim", c_int), ("vocab", c_int), ("rms_eps", c_float), ("mrope_sec0", c_int), ("mrope_sec1", c_int),
                ("attn_scale", c_float), ("rope_qk_scale", c_float), ("rope_long_from", c_int)]


class LlmLayer(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "wqkv", "bqkv", "wo", "ln2_w", "wgu", "wdown",
                                        "wqkv_sb", "wo_sb", "wgu_sb", "wdown_sb")]


class LlmGlobals(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("embed", "final_norm_w", "lm_head", "inv_freq", "embed_sb", "lm_head_sb")]


class KvPool(C.Structure):
    _fields_ = [("kpool", c_void_p), ("vpool", c_void_p), ("layer_stride", c_size_t), ("block_table", c_void_p),
                ("max_pages", c_int), ("kpool8", c_void_p), ("vpool8", c_void_p), ("ksb", c_void_p), ("vsb", c_void_p),
                ("q8_skip_last", c_int)]


class PrefillArgs(C.Structure):
    _fields_ = [("h", c_void_p), ("T", c_int), ("pos_t", c_void_p), ("pos_h", c_void_p), ("pos_w", c_void_p),
                ("kv_seq", c_void_p), ("kv_slot", c_void_p), ("cu_seqlens", c_void_p), ("nseg", c_int),
                ("total_qblocks", c_int), ("xn", c_void_p), ("qkv", c_void_p), ("attn", c_void_p), ("act", c_void_p),
                ("last_rows", c_void_p), ("n_last", c_int), ("xlast", c_void_p), ("logits", c_void_p), ("rope_long", c_int)]


class EncLayer(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2")]


class PenaltyArgs(C.Structure):
    _fields_ = [("hist", c_void_p), ("hist_len", c_void_p), ("hist_cap", c_int), ("rep_penalty", c_float), ("rep_ctx", c_int),
                ("pres_penalty", c_float), ("pres_ctx", c_int), ("freq_penalty", c_float), ("freq_ctx", c_int),
                ("bias_idx", c_void_p), ("bias_val", c_void_p), ("n_bias", c_int), ("row_params", c_void_p),
                ("bias_stride", c_int)]


class DecodeArgs(C.Structure):
    _fields_ = [("B", c_int), ("tok", c_void_p), ("pos", c_void_p), ("ctx", c_void_p), ("step", c_void_p),
                ("h", c_void_p), ("qkv", c_void_p), ("attn", c_void_p), ("act", c_void_p), ("logits", c_void_p),
                ("logprobs", c_void_p), ("scratch", c_void_p), ("part_o", c_void_p), ("part_ml", c_void_p),
                ("sample_ws", c_void_p), ("out_ring", c_void_p), ("ring_len", c_int), ("nsplit", c_int),
                ("temperature", c_float), ("top_p", c_float), ("min_p", c_float), ("top_k", c_int), ("seed", c_uint),
                ("flags", c_int), ("penalties", C.POINTER(PenaltyArgs))]


class SamplerParams(C.Structure):
    """vlm_sampler_params (include/vlm_hip.h): make_sampler's arguments as C doubles / ints"""
    _fields_ = [("temperature", C.c_double), ("top_p", C.c_double), ("min_p", C.c_double), ("min_tokens_to_keep", c_int),
                ("top_k", c_int), ("top_n_sigma", C.c_double), ("p_less", c_int), ("typical_p", C.c_double),
                ("xtc_probability", C.c_double), ("xtc_threshold", C.c_double), ("xtc_special_tokens", c_void_p),
                ("n_xtc_special", c_int), ("sort_workspace", c_void_p), ("seed", c_uint), ("input_is_logprobs", c_int)]


class VitConfig(C.Structure):
    _fields_ = [("depth", c_int), ("embed_dim", c_int), ("n_heads", c_int), ("mlp_hidden", c_int), ("patch_k", c_int),
                ("merge", c_int), ("out_dim", c_int), ("ln_eps", c_float), ("qk_interleaved", c_int)]


class VitBlock(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "wqkv", "bqkv", "wproj", "bproj", "ln2_w", "ln2_b", "wfc1",
                                        "bfc1", "wfc2", "bfc2")]


class VitGlobals(C.Structure):
    _fields_ = [(n, c_void_p) for n in ("wpatch", "ln_q_w", "ln_q_b", "wm0", "bm0", "wm2", "bm2")]


class VitArgs(C.Structure):
    _fields_ = [("patches", c_void_p), ("N", c_int), ("cos_tab", c_void_p), ("sin_tab", c_void_p),
                ("cu_seqlens", c_void_p), ("nseg", c_int), ("total_qblocks", c_int), ("x", c_void_p), ("xn", c_void_p),
                ("qkv", c_void_p), ("attn", c_void_p), ("mlp", c_void_p), ("mrg", c_void_p), ("out", c_void_p),
                ("uniform_segments", c_int)]


DECODE_FUSED_TAIL = 1                       # vlm_decode_args.flags
DECODE_ACT16 = 2                            # `act` holds 16 x intermediate_size elements (tiled hand-over to the down projection)
TUNE_MFMA_GEMV, TUNE_ATTN_PAGESPLIT, TUNE_GEMV_VARIANT, TUNE_ATTN_MERGE = 6, 7, 8, 9  # vlm_llm_set_tuning keys (include/vlm_hip.h)

P = C.POINTER
# name -> (restype, argtypes); every symbol include/vlm_hip.h declares
SIGNATURES = {
    "vlm_abi_version": (c_int, []),
    "vlm_gemm_bf16": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "vlm_gemm_bf16_rope2d": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "vlm_gemm_set_staging": (c_int, [c_int]),
    "vlm_gemv_bf16": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_float, c_int, c_void_p]),
    "vlm_gemv_qkv_rope_kvwrite": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p] + [c_int] * 6
                                  + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p]),
    "vlm_gemv_workspace_bytes": (C.c_size_t, []),
    "vlm_gemv_bf16_ws": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_float, c_int, c_void_p, c_void_p]),
    "vlm_gemv_qkv_rope_kvwrite_ws": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p] + [c_int] * 6
                                     + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlm_gemv_attn_out": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "vlm_gemv_w4": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_float, c_int, c_void_p]),
    "vlm_gemv_w4_qkv_rope_kvwrite": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6
                                     + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p]),
    "vlm_gemv_w4_ws": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_float, c_int, c_void_p, c_void_p]),
    "vlm_gemv_w4_qkv_rope_kvwrite_ws": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6
                                        + [c_void_p] * 4 + [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vlm_dequant_w4": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "vlm_layernorm": (c_int, [c_void_p] * 4 + [c_int, c_int, c_float, c_void_p]),
    "vlm_rmsnorm_residual": (c_int, [c_void_p] * 5 + [c_int, c_int, c_float, c_void_p]),
    "vlm_rope2d_vision": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "vlm_mrope_kvwrite": (c_int, [c_void_p] + [c_int] * 5 + [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 3 + [c_int]
                          + [c_void_p] * 3),
    "vlm_mrope_kvwrite_scaled": (c_int, [c_void_p] + [c_int] * 5 + [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 3 + [c_int]
                                 + [c_void_p] * 2 + [c_float, c_void_p]),
    "vlm_kv_gather": (c_int, [c_void_p] + [c_int] * 5 + [c_void_p] * 3 + [c_int] + [c_void_p] * 3),
    "vlm_attn_prefill": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p] + [c_int] * 5 + [c_float, c_int, c_void_p]),
    "vlm_attn_decode_paged": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 5
                              + [c_float, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vlm_attn_decode_paged_split": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 5
                                    + [c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vlm_encoder_forward": (c_int, [c_void_p, c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_float, c_int, c_void_p, c_int, c_int,
                                                                                          c_float, c_int, c_void_p]),
    "vlm_gemm_w4": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "vlm_kv_quantize_tokens": (c_int, [c_void_p] * 6 + [c_size_t, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                       c_void_p]),
    "vlm_attn_decode_paged_q8": (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_int, c_void_p] + [c_int] * 5
                                 + [c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vlm_gemv_attn_out_bf16": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "vlm_embed_gather": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "vlm_scatter_image_rows": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "vlm_cast_f32_bf16_pad": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "vlm_decode_advance": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_int, c_void_p]),
    "vlm_sample_workspace_bytes": (c_size_t, [c_int]),
    "vlm_sample": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float,
                           c_float, c_float, c_int, c_uint, c_void_p, c_void_p]),
    "vlm_kv_move_tokens": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                   c_int, c_int, c_void_p]),
    "vlm_kv_append_tokens": (c_int, [c_void_p] * 4 + [c_int] + [C.c_long] * 4 + [c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vlm_sample_sort_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vlm_sample_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                              C.POINTER(SamplerParams), c_void_p, c_void_p]),
    "vlm_apply_logit_penalties": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, C.POINTER(PenaltyArgs), c_void_p]),
    "vlm_sample_greedy_advance": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "vlm_llm_set_tuning": (c_int, [c_void_p, c_int, c_int]),
    "vlm_llm_get_tuning": (c_int, [c_void_p, c_int]),
    "vlm_llm_create": (c_int, [P(LlmConfig), P(c_void_p)]),
    "vlm_llm_destroy": (c_int, [c_void_p]),
    "vlm_llm_set_layer": (c_int, [c_void_p, c_int, P(LlmLayer)]),
    "vlm_llm_set_globals": (c_int, [c_void_p, P(LlmGlobals)]),
    "vlm_llm_set_kv": (c_int, [c_void_p, P(KvPool)]),
    "vlm_llm_prefill": (c_int, [c_void_p, P(PrefillArgs), c_void_p]),
    "vlm_llm_decode_step": (c_int, [c_void_p, P(DecodeArgs), c_void_p]),
    "vlm_llm_decode_forward": (c_int, [c_void_p, P(DecodeArgs), c_void_p]),
    "vlm_llm_decode_graph_build": (c_int, [c_void_p, P(DecodeArgs), c_void_p]),
    "vlm_llm_decode_graph_launch": (c_int, [c_void_p, c_void_p]),
    "vlm_llm_decode_launches": (c_int, [c_void_p]),
    "vlm_vit_create": (c_int, [P(VitConfig), P(c_void_p)]),
    "vlm_vit_destroy": (c_int, [c_void_p]),
    "vlm_vit_set_block": (c_int, [c_void_p, c_int, P(VitBlock)]),
    "vlm_vit_set_globals": (c_int, [c_void_p, P(VitGlobals)]),
    "vlm_vit_forward": (c_int, [c_void_p, P(VitArgs), c_void_p]),
}

_lib = None


class VlmHipError(RuntimeError):
    pass


def lib():
    """Load libvlm_hip.so once; raise loudly when it is absent (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VlmHipError(
                f"{LIB_PATH} not found - the HIP operator library is required (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `python mlx-vlm_amd/build.py`.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        kind = {1: "bad argument", 2: "unsupported shape"}.get(rc, f"HIP error {rc - 1000}" if rc >= 1000 else "error")
        raise VlmHipError(f"libvlm_hip {what} failed: rc={rc} ({kind})")


class _PinnedRing:
    """One pinned staging buffer for every small host -> device upload of the process.  `Tensor.pin_memory()` per
    upload goes through torch's pinned-memory allocator: a size class it has no idle block for costs a
    hipHostMalloc (tens of ms for MB-sized blocks - measured as sporadic 60 ms stalls inside the prefill enqueue).
    Regions are handed out in address order and wrap around; a region is reused only after the copy that read it
    has executed (event per upload)."""

    def __init__(self, nbytes: int = 64 << 20):
        import collections

        import torch

        self.buf = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        self.head = 0
        self.inflight = collections.deque()           # (start, end, event), oldest first

    def stage(self, t, device, out=None):
        """out: device tensor (or slice) of t's shape / dtype to fill instead of allocating"""
        import torch

        n = t.numel() * t.element_size()
        if n == 0 or n > self.buf.numel() // 2:       # rare and large: plain (synchronising) copy
            return t.to(device) if out is None else out.copy_(t)
        start = (self.head + 255) & ~255
        if start + n > self.buf.numel():
            start = 0
        end = start + n
        # the ring has come around: every upload still reading bytes of [start, end) must have executed.  Entries are in
        # issue order and uploads of one stream complete in that order, so waiting for the NEWEST overlapping one covers
        # the older ones; the whole deque is scanned (after a wrap the head may sit in the skipped tail of the buffer,
        # not overlap and still be pending, with overlapping entries behind it)
        newest = -1
        for i, (s0, e0, _) in enumerate(self.inflight):
            if s0 < end and start < e0:
                newest = i
        if newest >= 0:
            self.inflight[newest][2].synchronize()
            for _ in range(newest + 1):
                self.inflight.popleft()
        while self.inflight and self.inflight[0][2].query():
            self.inflight.popleft()
        view = self.buf[start:end].view(t.dtype).view(t.shape)
        # plain single-threaded memmove, NOT view.copy_(t): torch's CPU copy fans out over every core it sees (128 on the
        # MI355X hosts) and the OpenMP workers keep spinning after the region; inside a CPU-quota'd container that burns
        # the cgroup's budget and the whole process is throttled for the rest of the 100 ms scheduler period - measured
        # as single 80-90 ms "memcpy" calls of 2.7 MB in the continuous-batching loop (profiles/r02_continuous_diag.txt)
        if not t.is_contiguous():
            t = t.contiguous()
        C.memmove(self.buf.data_ptr() + start, t.data_ptr(), n)
        out = view.to(device, non_blocking=True) if out is None else out.copy_(view, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.inflight.append((start, end, ev))
        self.head = end
        return out


_ring = None


def h2d(x, device):
    """Host array / CPU tensor -> device, staged through pinned memory and enqueued asynchronously on the current stream.
    A copy from pageable memory makes the host wait for the stream to drain first: with it, the host cannot enqueue the
    LLM prefill while the ViT runs, and an admission on a side stream would stall the decode loop at every small index
    upload."""
    global _ring
    import numpy as np
    import torch

    t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
    if t.device.type != "cpu":
        return t.to(device)
    if not torch.cuda.is_available() or torch.device(device).type == "cpu":
        return t.to(device)
    if _ring is None:
        _ring = _PinnedRing()
    return _ring.stage(t.contiguous(), device)


def h2d_cat(parts, device):
    """dim-0 concatenation of host tensors assembled ON the device: each part is staged through the pinned ring into its
    slice of one device tensor.  A host-side torch.cat of the pixel rows of an admission (2.7 MB of fp32 per 336 x 336
    image) is a pass over freshly faulted pageable memory - measured 4 ms per call inside the continuous-batching loop."""
    global _ring
    import numpy as np
    import torch

    ts = [torch.from_numpy(np.ascontiguousarray(p)) if isinstance(p, np.ndarray) else torch.as_tensor(p) for p in parts]
    if len(ts) == 1:
        return h2d(ts[0], device)
    if any(t.device.type != "cpu" for t in ts) or not torch.cuda.is_available() or torch.device(device).type == "cpu" \
            or any(t.dtype != ts[0].dtype or t.shape[1:] != ts[0].shape[1:] for t in ts):
        return torch.cat([t.to(device) for t in ts], dim=0)
    if _ring is None:
        _ring = _PinnedRing()
    out = torch.empty((sum(t.shape[0] for t in ts),) + tuple(ts[0].shape[1:]), dtype=ts[0].dtype, device=device)
    off = 0
    for t in ts:
        _ring.stage(t.contiguous(), device, out=out[off:off + t.shape[0]])
        off += t.shape[0]
    return out
