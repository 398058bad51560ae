"""Qwen2-VL processor (host, numpy) - mirror of the reference's
mlx_vlm/models/qwen2_vl/processing_qwen2_vl.py:62-127 (placeholder expansion +
tokenise) and of the numpy image processor it uses,
mlx_vlm/models/qwen3_vl/processing_qwen3_vl.py:182-205 (smart resize),
:302-354 (_process_one: bicubic resize, rescale, normalise, duplicate the frame
along T, 10-D reshape/transpose into merge-window patch order)."""
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


def load_image(img) -> np.ndarray:
    """-> uint8 [C, H, W] RGB (reference processing_qwen3_vl.py `_to_numpy_image`)."""
    from PIL import Image

    if isinstance(img, str):
        img = Image.open(img)
    if hasattr(img, "convert"):
        arr = np.array(img.convert("RGB"))
    else:
        arr = np.asarray(img)
    if arr.ndim == 2:
        arr = np.stack([arr] * 3, axis=-1)
    if arr.ndim == 3 and arr.shape[-1] in (1, 3, 4):
        arr = np.transpose(arr, (2, 0, 1))
    if arr.shape[0] == 4:
        arr = arr[:3]
    return arr


class Qwen2VLImageProcessor:
    model_input_names = ["pixel_values", "image_grid_thw"]

    def __init__(self, patch_size: int = 14, temporal_patch_size: int = 2, merge_size: int = 2,
                 min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280, do_rescale: bool = True,
                 rescale_factor: float = 1 / 255.0, do_normalize: bool = True, image_mean: Optional[List[float]] = None,
                 image_std: Optional[List[float]] = None, **kwargs):
        self.patch_size, self.temporal_patch_size, self.merge_size = patch_size, temporal_patch_size, merge_size
        self.min_pixels, self.max_pixels = min_pixels, max_pixels
        self.do_rescale, self.rescale_factor, self.do_normalize = do_rescale, rescale_factor, do_normalize
        self.image_mean = image_mean or [0.5, 0.5, 0.5]
        self.image_std = image_std or [0.5, 0.5, 0.5]

    def _process_one(self, image: np.ndarray):
        from PIL import Image

        C, H, W = image.shape
        rh, rw = smart_resize(H, W, self.patch_size * self.merge_size, self.min_pixels, self.max_pixels)
        frame = image
        if (H, W) != (rh, rw):
            pil = Image.fromarray(np.transpose(image, (1, 2, 0))).resize((rw, rh), resample=Image.BICUBIC)
            frame = np.transpose(np.array(pil), (2, 0, 1))
        ps, tps, ms = self.patch_size, self.temporal_patch_size, self.merge_size
        gh, gw = rh // ps, rw // ps
        if frame.dtype == np.uint8 and image.dtype == np.uint8:
            return self._patchify_u8(frame, gh, gw), [1, gh, gw]
        img = self._normalise(frame.astype(np.float32), rescale=self.do_rescale and image.dtype == np.uint8)
        patches = np.repeat(img[None, None, ...], tps, axis=1)
        patches = patches.reshape(1, 1, tps, C, gh // ms, ms, ps, gw // ms, ms, ps)
        patches = patches.transpose(0, 1, 4, 7, 5, 8, 3, 2, 6, 9)
        return patches.reshape(gh * gw, C * tps * ps * ps), [1, gh, gw]

    def _normalise(self, img: np.ndarray, rescale: bool) -> np.ndarray:
        """float32 [C, ...]: x * rescale_factor, then (x - mean) / std, each a float32 numpy op (the reference's order,
        processing_qwen3_vl.py:302-331)."""
        if rescale:
            img = img * np.float32(self.rescale_factor)
        if self.do_normalize:
            shape = (-1,) + (1,) * (img.ndim - 1)
            mean = np.array(self.image_mean, dtype=np.float32).reshape(shape)
            std = np.array(self.image_std, dtype=np.float32).reshape(shape)
            img = (img - mean) / std
        return img

    def _patchify_u8(self, frame: np.ndarray, gh: int, gw: int) -> np.ndarray:
        """uint8 [C, H, W] -> float32 [gh * gw, C * T * ps * ps], bit-identical to the float path above and several times faster:
        rescale + normalise is a pure function of (channel, byte), so it is a C x 256 table built with the very same
        float32 operations; the patch shuffle (rows (gh/m, gw/m, m, m), columns (C, T, ph, pw)) is done on the bytes,
        a quarter of the traffic, and the table expands them straight into both temporal copies of the output."""
        C = frame.shape[0]
        ps, tps, ms = self.patch_size, self.temporal_patch_size, self.merge_size
        lut = self._normalise(np.broadcast_to(np.arange(256, dtype=np.float32), (C, 256)).copy(), rescale=self.do_rescale)
        out = np.empty((gh // ms, gw // ms, ms, ms, C, tps, ps, ps), dtype=np.float32)
        blocks = frame.reshape(C, gh // ms, ms, ps, gw // ms, ms, ps)
        for c in range(C):
            values = lut[c].take(blocks[c].transpose(0, 3, 1, 4, 2, 5))   # bytes in patch order -> [gh/m, gw/m, m, m, ps, ps]
            for t in range(tps):
                out[:, :, :, :, c, t] = values
        return out.reshape(gh * gw, C * tps * ps * ps)

    def __call__(self, images, **kwargs):
        ps, thw = [], []
        for im in images:
            p, g = self._process_one(im if isinstance(im, np.ndarray) and im.ndim == 3 else load_image(im))
            ps.append(p)
            thw.append(g)
        pixel_values = ps[0] if len(ps) == 1 else np.concatenate(ps, axis=0)     # no second 2.7 MB copy for one image
        return {"pixel_values": pixel_values, "image_grid_thw": np.array(thw, dtype=np.int64)}

    def num_image_tokens(self, height: int, width: int) -> int:
        rh, rw = smart_resize(height, width, self.patch_size * self.merge_size, self.min_pixels, self.max_pixels)
        return (rh // self.patch_size) * (rw // self.patch_size) // self.merge_size ** 2


class Qwen2VLProcessor:
    """reference processing_qwen2_vl.py:62-127: expand each <|image_pad|> to grid.prod() // merge^2 copies, tokenise."""

    def __init__(self, image_processor, tokenizer, image_token: str = "<|image_pad|>"):
        self.image_processor = image_processor
        self.tokenizer = tokenizer
        self.image_token = getattr(tokenizer, "image_token", image_token)

    def __call__(self, images=None, text=None, **kwargs):
        image_inputs = {}
        if images is not None:
            image_inputs = self.image_processor(images)
            grid = image_inputs["image_grid_thw"]
        if not isinstance(text, list):
            text = [text]
        text = list(text)
        if images is not None:
            merge_length = self.image_processor.merge_size ** 2
            index = 0
            for i in range(len(text)):
                while self.image_token in text[i]:
                    n = int(grid[index].prod()) // merge_length
                    text[i] = text[i].replace(self.image_token, "<|placeholder|>" * n, 1)
                    index += 1
                text[i] = text[i].replace("<|placeholder|>", self.image_token)
        enc = self.tokenizer(text, **kwargs)
        out = {"input_ids": np.asarray(enc["input_ids"], dtype=np.int64)}
        if "attention_mask" in enc:
            out["attention_mask"] = np.asarray(enc["attention_mask"], dtype=np.int64)
        out.update(image_inputs)
        return out

    def decode(self, *a, **k):
        return self.tokenizer.decode(*a, **k)

    def batch_decode(self, *a, **k):
        return self.tokenizer.batch_decode(*a, **k)
