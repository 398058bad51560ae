"""Oracle restatement of the Qwen2-VL image processor (TEST INFRASTRUCTURE).

Follows /root/reference/mlx_vlm/models/qwen3_vl/processing_qwen3_vl.py:182-205
(_smart_resize_image) and :302-354 (Qwen3VLImageProcessor._process_one), which
is the processor Qwen2-VL uses (models/qwen2_vl/processing_qwen2_vl.py:144-168,
patch_size 14, merge 2 -> factor 28).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280) -> Tuple[int, int]:
    if max(height, width) / min(height, width) > 200:
        raise ValueError(
            f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def resize_bicubic(chw_u8: np.ndarray, h: int, w: int) -> np.ndarray:
    """PIL bicubic resize per frame (processing_qwen3_vl.py `_resize_video_frames`)."""
    from PIL import Image

    C, H, W = chw_u8.shape
    if (H, W) == (h, w):
        return chw_u8
    img = Image.fromarray(np.transpose(chw_u8, (1, 2, 0)))
    img = img.resize((w, h), resample=Image.BICUBIC)
    return np.transpose(np.array(img), (2, 0, 1))


def process_one(image_chw_u8: np.ndarray, patch_size: int = 14, temporal_patch_size: int = 2,
                merge_size: int = 2, image_mean: Optional[List[float]] = None,
                image_std: Optional[List[float]] = None, min_pixels: int = 56 * 56,
                max_pixels: int = 14 * 14 * 4 * 1280):
    """-> (pixel_values f32 [gh*gw, C*tps*ps*ps], [1, gh, gw])."""
    C, H, W = image_chw_u8.shape
    rh, rw = smart_resize(H, W, patch_size * merge_size, min_pixels, max_pixels)
    frame = resize_bicubic(image_chw_u8, rh, rw)
    img = frame.astype(np.float32)
    if image_chw_u8.dtype == np.uint8:
        img = img * np.float32(1 / 255.0)
    mean = np.array(image_mean or [0.5, 0.5, 0.5], dtype=np.float32)[:, None, None]
    std = np.array(image_std or [0.5, 0.5, 0.5], dtype=np.float32)[:, None, None]
    img = (img - mean) / std
    patches = np.repeat(img[None, None, ...], temporal_patch_size, axis=1)
    ps, tps, ms = patch_size, temporal_patch_size, merge_size
    gh, gw = rh // ps, rw // ps
    patches = patches.reshape(1, 1, tps, C, gh // ms, ms, ps, gw // ms, ms, ps)
    patches = patches.transpose(0, 1, 4, 7, 5, 8, 3, 2, 6, 9)
    flat = patches.reshape(gh * gw, C * tps * ps * ps)
    return flat, [1, gh, gw]


def process(images: List[np.ndarray], **kw):
    ps, thw = [], []
    for im in images:
        p, g = process_one(im, **kw)
        ps.append(p)
        thw.append(g)
    return np.concatenate(ps, axis=0), np.array(thw, dtype=np.int64)


def expand_image_placeholders(token_ids: List[int], image_token_id: int, grid_thw: np.ndarray,
                              merge_size: int = 2) -> List[int]:
    """Qwen2VLProcessor.__call__ placeholder expansion
    (processing_qwen2_vl.py:93-105) on token ids: the i-th image token is
    replaced by grid.prod() // merge^2 copies."""
    out, idx = [], 0
    for t in token_ids:
        if t == image_token_id:
            n = int(np.prod(grid_thw[idx])) // (merge_size ** 2)
            out += [image_token_id] * n
            idx += 1
        else:
            out.append(t)
    return out
