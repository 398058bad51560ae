"""Image processor and prompt assembly for nanoLLaVA - host mirror of the reference's `ImageProcessor`
(`mlx_vlm/models/llava_bunny/llava_bunny.py:24-57` over `models/base.py:121-194`) and of the `<image>` branch of
`prepare_inputs` (`utils.py:2064-2095`)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

IMAGE_TOKEN_INDEX = -200


class ImageProcessor:
    """RGB -> resize to `size` (PIL bicubic) -> x * (1 / 255) -> (x - mean) / std -> channels first, float32."""

    def __init__(self, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), size=(384, 384), rescale_factor=1 / 255):
        self.image_mean, self.image_std, self.size, self.rescale_factor = image_mean, image_std, tuple(size), rescale_factor

    def preprocess(self, images) -> List[np.ndarray]:
        from PIL import Image

        if not isinstance(images, (list, tuple)):
            images = [images]
        # rescale (float64 product cast to float32, as transformers' `rescale`) and normalise are pure functions of
        # (channel, byte): one 3 x 256 table built with exactly those operations, expanded per channel
        mean = np.asarray(self.image_mean, dtype=np.float32)
        std = np.asarray(self.image_std, dtype=np.float32)
        levels = (np.arange(256, dtype=np.float64) * self.rescale_factor).astype(np.float32)
        lut = (levels[None, :] - mean[:, None]) / std[:, None]                     # float32 [3, 256]
        out = []
        for img in images:
            if not isinstance(img, Image.Image):
                arr = np.asarray(img)
                if arr.ndim == 3 and arr.shape[0] in (1, 3) and arr.shape[-1] not in (1, 3):
                    arr = np.transpose(arr, (1, 2, 0))                 # channels first input
                img = Image.fromarray(arr.astype(np.uint8))
            pil = img.convert("RGB").resize((self.size[1], self.size[0]), resample=Image.BICUBIC)
            hwc = np.asarray(pil)
            out.append(np.stack([lut[c].take(hwc[:, :, c]) for c in range(3)]))
        return out

    __call__ = preprocess


def assemble_input_ids(tokenize, prompts: Sequence[str], pad_token_id: int, image_token_index: int = IMAGE_TOKEN_INDEX):
    """`prepare_inputs` for BaseImageProcessor models (utils.py:2064-2095): every prompt is split at "<image>", the
    chunks are tokenised separately and joined as chunk0 + [image_token_index] + chunk1, rows right-padded to the
    longest.  `tokenize(str) -> list of ids`.  -> (input_ids int64 [B, L], attention_mask int32 [B, L])"""
    rows = []
    for prompt in prompts:
        chunks = [list(tokenize(c)) for c in prompt.split("<image>")]
        if len(chunks) < 2:
            raise ValueError('the prompt needs an "<image>" placeholder')
        rows.append(chunks[0] + [image_token_index] + chunks[1])
    L = max(len(r) for r in rows)
    ids = np.array([r + [pad_token_id] * (L - len(r)) for r in rows], dtype=np.int64)
    return ids, (ids != pad_token_id).astype(np.int32)
