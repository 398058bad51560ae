"""Vision side of Phi-3.5-vision on the C-ABI kernels - host mirror of the reference's `mlx_vlm/models/phi3_v/vision.py`:
CLIP ViT-L/14-336 (VisionEmbeddings 113-148: bias-free Conv2d patch embed, class token, learned positions; ClipModel
151-176: `pre_layrnorm`, pre-LN encoder layers 83-102 with separate q / k / v / out projections and FastGELUMLP), the HD
transform with its `sub_GN` / `glb_GN` separator rows and the `img_projection` MLP (VisionModel.__call__ 207-262).

What runs where:
  * per view (336 x 336 -> 577 tokens): patch GEMM with the position rows as the residual operand (the class row
    `cls + pos[0]` is one typed add done at load time), LayerNorm, then per layer [LayerNorm -> fused q|k|v GEMM + bias ->
    flash attention over each view's 577 tokens (hd 64, `cu_seqlens`) -> out GEMM + bias + residual -> LayerNorm -> fc1 GEMM
    + bias + quick-GELU -> fc2 GEMM + bias + residual];
  * the model reads `encoder_states[-2]` (vision.py:224-226): the LAST encoder layer and `post_layernorm` cannot
    influence any output and are neither loaded onto the device nor run;
  * the HD row assembly (2 x 2 merge of the 24 x 24 grid, the reference's plain reshape of the local views, separator
    columns) is data movement on device tensors; the two projection Linears are GEMMs with bias (+ erf-GELU) epilogues.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from ... import _lib, ops
from .config import VisionConfig


class VisionModel:
    def __init__(self, config, device="cuda"):
        """`config`: the model's ModelConfig (the reference passes it whole and reads `hidden_size` for the projection);
        the tower's own dimensions come from `config.vision_config` (CLIP ViT-L/14-336 unless a test shrinks it)."""
        self.model_config = config
        c = config.vision_config if hasattr(config, "vision_config") else VisionConfig()
        self.config = c
        self.model_type = "phi3_v"
        self.device = device
        self.grid = c.image_size // c.patch_size
        self.num_patches = self.grid * self.grid
        self.head_dim = c.hidden_size // c.num_attention_heads
        if self.head_dim not in (64, 80, 128):
            raise NotImplementedError(f"vision head_dim {self.head_dim}")
        self.image_dim_out = c.hidden_size
        self.patch_dim = c.patch_size * c.patch_size * c.num_channels
        self.patch_k = (self.patch_dim + 63) // 64 * 64
        self.n_run_layers = c.num_hidden_layers - 1             # encoder_states[-2]
        self._w: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ weights
    def load_weights(self, W: Dict[str, torch.Tensor]):
        """W: names relative to `model.vision_embed_tokens.` (`img_processor.vision_model.*`, `glb_GN`, `sub_GN`,
        `img_projection.{0,2}.*`); patch weight (O, kH, kW, C) as `sanitize` leaves it.  A quantized `img_projection`
        (`.scales` present) is dequantized once: the projection runs on the bf16 GEMM."""
        self._enc = None            # (the native layer loop's weight table is rebuilt on first use)
        from .. import quantized as Qz

        c, dev, bf = self.config, self.device, torch.bfloat16
        E = c.hidden_size
        P = "img_processor.vision_model."

        def g(name):
            return W[name].to(device=dev, dtype=bf).contiguous()

        def lin_w(path):
            if Qz.has_scales(W, path):
                return Qz.dequantize_bf16(Qz.take(W, path), dev)
            return g(path + ".weight")

        wp = torch.zeros(E, self.patch_k, dtype=bf, device=dev)
        wp[:, : self.patch_dim] = g(P + "embeddings.patch_embedding.weight").reshape(E, -1)
        pos = g(P + "embeddings.position_embedding.weight")
        cls = g(P + "embeddings.class_embedding").reshape(1, E)
        self._w.update(wpatch=wp, pos_patches=pos[1:].contiguous(),
                       cls_row=(cls.float() + pos[:1].float()).to(bf),          # the typed add of the class row, once
                       pre_w=g(P + "pre_layrnorm.weight"), pre_b=g(P + "pre_layrnorm.bias"))
        for i in range(self.n_run_layers):
            p = f"{P}encoder.layers.{i}."
            self._w.update({
                f"{i}.wqkv": torch.cat([lin_w(p + f"self_attn.{n}") for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous(),
                f"{i}.bqkv": torch.cat([g(p + f"self_attn.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous(),
                f"{i}.wo": lin_w(p + "self_attn.out_proj"), f"{i}.bo": g(p + "self_attn.out_proj.bias"),
                f"{i}.ln1w": g(p + "layer_norm1.weight"), f"{i}.ln1b": g(p + "layer_norm1.bias"),
                f"{i}.ln2w": g(p + "layer_norm2.weight"), f"{i}.ln2b": g(p + "layer_norm2.bias"),
                f"{i}.w1": lin_w(p + "mlp.fc1"), f"{i}.b1": g(p + "mlp.fc1.bias"),
                f"{i}.w2": lin_w(p + "mlp.fc2"), f"{i}.b2": g(p + "mlp.fc2.bias")})
        self._w.update(glb_GN=g("glb_GN").reshape(1, 4 * E), sub_GN=g("sub_GN").reshape(1, 4 * E),
                       p0w=lin_w("img_projection.0"), p0b=g("img_projection.0.bias"),
                       p2w=lin_w("img_projection.2"), p2b=g("img_projection.2.bias"))
        return self

    # ------------------------------------------------------------------ CLIP tower
    def patchify(self, views: torch.Tensor) -> torch.Tensor:
        """[N, 3, 336, 336] float -> bf16 [N * 576, patch_k], a row = one patch flattened (kH, kW, C)-major (what the
        NHWC Conv2d with kernel = stride = patch contracts)."""
        c, G, P = self.config, self.grid, self.config.patch_size
        x = views if views.is_cuda else _lib.h2d(views, self.device)
        N = x.shape[0]
        x = x.to(torch.float32).reshape(N, c.num_channels, G, P, G, P)
        x = x.permute(0, 2, 4, 3, 5, 1).reshape(N * G * G, self.patch_dim).contiguous()      # data movement only
        return ops.cast_pad(x, self.patch_k)

    def clip_features(self, views: torch.Tensor) -> torch.Tensor:
        """views [N, 3, 336, 336] -> bf16 [N, 576, E]: `encoder_states[-2][:, 1:]`"""
        c, w = self.config, self._w
        E, H, hd, Np = c.hidden_size, c.num_attention_heads, self.head_dim, self.num_patches
        L = Np + 1
        patches = self.patchify(views)
        N = patches.shape[0] // Np
        x = torch.empty(N * L, E, dtype=torch.bfloat16, device=self.device)
        x3 = x.view(N, L, E)
        x3[:, 0] = w["cls_row"]
        for n in range(N):            # conv rows + position rows (the table is the residual operand of each view)
            ops.gemm(patches[n * Np:(n + 1) * Np], w["wpatch"], res=w["pos_patches"], out=x[n * L + 1:(n + 1) * L],
                     epilogue=ops.EPI_RESIDUAL)
        xn = torch.empty_like(x)
        ops.layernorm(x, w["pre_w"], w["pre_b"], 1e-5, out=xn)       # nn.LayerNorm(hidden): MLX's default eps
        x, xn = xn, x
        cu = _lib.h2d(np.arange(N + 1, dtype=np.int32) * L, self.device)
        nqb = N * ((L + 127) // 128)
        scale = float(hd) ** -0.5
        # the encoder layers as ONE native call (vlm_encoder_forward: 7 launches per layer, no host work in between)
        if getattr(self, "_enc", None) is None:
            self._enc = ops.EncoderLayers(w, self.n_run_layers)
        self._enc.forward_(x, H, hd, c.layer_norm_eps, ops.EPI_GELU_FAST, cu, nqb, scale)
        return x.view(N, L, E)[:, 1:]

    # ------------------------------------------------------------------ HD transform (vision.py:228-256)
    def hd_rows(self, feat: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """feat [1 + h w (+ padding views), 576, C] of one image -> [(h w + 1) 144 + 1 + (h + 1) 12, 4 C]: local rows (each
        grid row closed by sub_GN), glb_GN, global rows.  Pure data movement."""
        C = feat.shape[-1]
        Hh = self.grid // 2
        sub_gn, glb_gn = self._w["sub_GN"], self._w["glb_GN"]

        def merged(x):
            n = x.shape[0]
            return x.reshape(n, Hh, 2, Hh, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(n, Hh, Hh, 4 * C)

        def with_sep(grid):
            R = grid.shape[1]
            return torch.cat([grid, sub_gn.reshape(1, 1, 1, 4 * C).expand(1, R, 1, 4 * C)], dim=2).reshape(-1, 4 * C)

        glb = with_sep(merged(feat[:1]))
        sub = with_sep(merged(feat[1: 1 + h * w]).reshape(1, h * Hh, w * Hh, 4 * C))     # the reference's plain reshape
        return torch.cat([sub, glb_gn, glb], dim=0).contiguous()

    def img_projection(self, rows: torch.Tensor) -> torch.Tensor:
        w = self._w
        hmid = ops.gemm(rows, w["p0w"], bias=w["p0b"], epilogue=ops.EPI_BIAS | ops.EPI_GELU_ERF)
        return ops.gemm(hmid, w["p2w"], bias=w["p2b"], epilogue=ops.EPI_BIAS)

    def image_features(self, pixel_values, image_sizes) -> List[torch.Tensor]:
        """pixel_values [B, T, 3, 336, 336], image_sizes [B, 2] (pixels) -> per image bf16 [cnt_b, hidden].  All views of
        all images go through the tower in one pass; the projections of all images in one GEMM pair."""
        pv = torch.as_tensor(pixel_values)
        B, T = pv.shape[:2]
        sizes = np.asarray(image_sizes.cpu() if isinstance(image_sizes, torch.Tensor) else image_sizes).reshape(B, 2)
        hw = [(int(sizes[b][0]) // 336, int(sizes[b][1]) // 336) for b in range(B)]
        # the views an image does not have (zero padding up to the batch maximum) never reach an output: not computed
        used = [b * T + t for b, (h, w) in enumerate(hw) for t in range(1 + h * w)]
        flat = pv.reshape(B * T, *pv.shape[2:])
        feat = self.clip_features(flat if len(used) == B * T else flat[used])
        rows, at = [], 0
        for h, w in hw:
            rows.append(self.hd_rows(feat[at: at + 1 + h * w], h, w))
            at += 1 + h * w
        proj = self.img_projection(torch.cat(rows, dim=0) if B > 1 else rows[0])
        return list(torch.split(proj, [r.shape[0] for r in rows], dim=0))

    def __call__(self, img_embeds, txt_embeds=None, img_sizes=None, positions=None, output_hidden_states=None):
        """reference signature (vision.py:207-214): writes the projected rows of image i into `txt_embeds` from the
        position of its first negative id on.  -> txt_embeds (modified in place, as the reference does)"""
        if output_hidden_states:
            raise NotImplementedError("hidden-state taps of the tower are outside the built path")
        idx = 0
        for rows in self.image_features(img_embeds, img_sizes):
            b, start = positions[idx]
            cnt = rows.shape[0]
            txt_embeds[b, start:start + cnt] = rows
            idx += cnt
        return txt_embeds

    # ------------------------------------------------------------------ checkpoint fix-ups (reference vision.py:264-280)
    def sanitize(self, weights):
        out = {}
        for k, v in weights.items():
            if "position_ids" in k:
                continue
            if "patch_embedding.weight" in k and v.ndim == 4:
                O, a, b_, _c = v.shape
                if not (O >= a and O >= b_ and a == b_):       # torch layout (O, C, kH, kW) -> (O, kH, kW, C)
                    v = v.permute(0, 2, 3, 1)
            out[k] = v
        return out
