"""Idefics2 connector on the C-ABI kernels - host mirror of the reference's `mlx_vlm/models/idefics2/idefics2.py:36-177`:
modality projection (silu-gated MLP) + perceiver resampler (n_latents learned queries; per layer RMSNorm of latents and of
the context, GQA attention of the latents over [context | latents] without mask or rope, o_proj + residual, RMSNorm,
gated MLP + residual; final RMSNorm).

The perceiver's cross-attention runs on the varlen flash-attention kernel: per image the rows [context | latents] form one
segment of a token-major buffer whose k / v columns hold the projections of every row and whose q columns hold the
latents' queries (context rows keep zero queries: their outputs are computed and discarded - 3 layers of a few hundred rows,
negligible next to the tower).  Heads are 96 wide, zero-padded to the kernel's 128 (zero rows in q / k / v, zero columns
in o_proj; the softmax scale stays 96 ** -0.5).  Gate / up rows are interleaved at load so silu(gate) * up is a GEMM epilogue."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from ... import _lib, ops

HEAD_PAD = 128


class Connector:
    def __init__(self, config, device="cuda"):
        self.config = config
        self.device = device
        self._w: Dict[str, torch.Tensor] = {}

    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: names relative to `connector.`"""
        p, dev, bf = self.config.perceiver_config, self.device, torch.bfloat16
        hd, H, Hkv = p.resampler_head_dim, p.resampler_n_heads, p.num_key_value_heads
        if hd > HEAD_PAD:
            raise NotImplementedError(f"resampler_head_dim {hd}")

        from .. import quantized as Qz

        def g(name):
            """a tensor, or - for a Linear the checkpoint holds in MLX 4 bits (`<path>.scales`, utils.py:961) - its bf16
            dequantisation: the connector runs on the bf16 GEMMs"""
            if name.endswith(".weight") and Qz.has_scales(W, name[: -len(".weight")]):
                return Qz.dequantize_bf16(Qz.take(W, name[: -len(".weight")]), dev).contiguous()
            return W[name].to(device=dev, dtype=bf).contiguous()

        def gated(prefix):
            gate, up = g(prefix + "gate_proj.weight"), g(prefix + "up_proj.weight")
            return torch.stack([gate, up], dim=1).reshape(2 * gate.shape[0], gate.shape[1]).contiguous()

        def pad_rows(w, heads):
            out = torch.zeros(heads, HEAD_PAD, w.shape[1], dtype=bf, device=dev)
            out[:, :hd] = w.reshape(heads, hd, -1)
            return out.reshape(heads * HEAD_PAD, -1)

        self._w.update(mp_gu=gated("modality_projection."), mp_down=g("modality_projection.down_proj.weight"),
                       latents=g("perceiver_resampler.latents"), norm=g("perceiver_resampler.norm.weight"))
        for i in range(p.resampler_depth):
            q = f"perceiver_resampler.layers.{i}."
            wo = g(q + "self_attn.o_proj.weight")
            wo_p = torch.zeros(wo.shape[0], H, HEAD_PAD, dtype=bf, device=dev)
            wo_p[:, :, :hd] = wo.reshape(wo.shape[0], H, hd)
            self._w.update({f"{i}.ln_lat": g(q + "input_latents_norm.weight"), f"{i}.ln_ctx": g(q + "input_context_norm.weight"),
                            f"{i}.wq": pad_rows(g(q + "self_attn.q_proj.weight"), H).contiguous(),
                            f"{i}.wkv": torch.cat([pad_rows(g(q + "self_attn.k_proj.weight"), Hkv),
                                                   pad_rows(g(q + "self_attn.v_proj.weight"), Hkv)], 0).contiguous(),
                            f"{i}.wo": wo_p.reshape(wo.shape[0], H * HEAD_PAD).contiguous(),
                            f"{i}.ln_post": g(q + "post_attention_layernorm.weight"),
                            f"{i}.gu": gated(q + "mlp."), f"{i}.down": g(q + "mlp.down_proj.weight")})
        return self

    def __call__(self, image_hidden: torch.Tensor, n_images: int) -> torch.Tensor:
        """image_hidden bf16 [n * L, E] (the tower's pooler output) -> [n * n_latents, D]"""
        cfg, w = self.config, self._w
        p, eps = cfg.perceiver_config, cfg.text_config.rms_norm_eps
        H, Hkv, nl = p.resampler_n_heads, p.num_key_value_heads, p.resampler_n_latents
        n = n_images
        L = image_hidden.shape[0] // n
        x = ops.gemm(ops.gemm(image_hidden, w["mp_gu"], epilogue=ops.EPI_SWIGLU), w["mp_down"])           # [n L, D]
        D = x.shape[1]
        S = L + nl
        h = w["latents"].repeat(n, 1)                                                                       # [n nl, D]
        QW, KVW = H * HEAD_PAD, 2 * Hkv * HEAD_PAD
        cu = _lib.h2d(np.arange(n + 1, dtype=np.int32) * S, self.device)
        nqb = n * ((S + 127) // 128)
        scale = float(p.resampler_head_dim) ** -0.5
        buf = torch.zeros(n * S, QW + KVW, dtype=torch.bfloat16, device=self.device)      # q | k | v per row of [context | latents]
        b3 = buf.view(n, S, QW + KVW)
        hs = torch.empty(n, S, D, dtype=torch.bfloat16, device=self.device)
        for i in range(p.resampler_depth):
            lat = ops.rmsnorm(h, w[f"{i}.ln_lat"], eps)
            ctx = ops.rmsnorm(x, w[f"{i}.ln_ctx"], eps)
            hs[:, :L] = ctx.view(n, L, D)
            hs[:, L:] = lat.view(n, nl, D)
            ops.gemm(hs.view(n * S, D), w[f"{i}.wkv"], out=buf[:, QW:])                                      # keys / values of every row
            b3[:, L:, :QW] = ops.gemm(lat, w[f"{i}.wq"]).view(n, nl, QW)                                    # queries: latent rows only
            o = ops.attn_prefill(buf[:, :QW], buf[:, QW: QW + Hkv * HEAD_PAD], buf[:, QW + Hkv * HEAD_PAD:], cu, nqb, H, Hkv,
                                 HEAD_PAD, scale, causal=False, uniform_segments=True)
            o_lat = o.view(n, S, QW)[:, L:].reshape(n * nl, QW)
            h = ops.gemm(o_lat, w[f"{i}.wo"], res=h, epilogue=ops.EPI_RESIDUAL)
            act = ops.gemm(ops.rmsnorm(h, w[f"{i}.ln_post"], eps), w[f"{i}.gu"], epilogue=ops.EPI_SWIGLU)
            h = ops.gemm(act, w[f"{i}.down"], res=h, epilogue=ops.EPI_RESIDUAL)
        return ops.rmsnorm(h, w["norm"], eps)
