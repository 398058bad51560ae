"""Image processor and prompt assembly of Phi-3.5-vision - host mirror of the reference's
`mlx_vlm/models/phi3_v/processing_phi3_v.py`: `_calc_hd_transform_size` / `_calc_padded_size` (78-110), `_hd_transform` /
`_pad_to_336` (113-138), `Phi3VImageProcessor` (141-291: global 336 x 336 view + row-major 336 x 336 tiles of the HD image,
CLIP mean / std, views zero-padded to the batch maximum, token-count rule) and `Phi3VProcessor` (294-535: every
`<|image_N|>` tag becomes `num_tokens(image N)` copies of the id -N between the separately tokenised text chunks).

Numerics of the pixel path (bit-exact against the reference, tests/test_oracle_ref_golden_phi3v.py): bytes -> float32
x / 255 -> (x - mean) / std in float64 (numpy promotes against the float64 constants) -> float32 when the result
becomes an array of the runtime.  A pure function of (channel, byte): one 3 x 256 table built with exactly those
operations."""
from __future__ import annotations

import math
import re
from typing import List, Sequence, Tuple

import numpy as np

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
TILE = 336


def calc_hd_transform_size(width: int, height: int, hd_num: int = 16) -> Tuple[int, int]:
    """-> (padded_width, padded_height): the largest scale x ceil(scale / ratio) tile grid within hd_num tiles"""
    transposed = width < height
    if transposed:
        width, height = height, width
    ratio = width / height
    scale = 1
    while scale * math.ceil(scale / ratio) <= hd_num:
        scale += 1
    scale -= 1
    new_w = int(scale * TILE)
    new_h = int(new_w / ratio)
    pw, ph = math.ceil(new_w / TILE) * TILE, math.ceil(new_h / TILE) * TILE
    return (ph, pw) if transposed else (pw, ph)


def _to_pil(img):
    from PIL import Image

    if isinstance(img, str):
        img = Image.open(img)
    if not isinstance(img, Image.Image):
        arr = np.asarray(img)
        if arr.ndim == 3 and arr.shape[0] in (1, 3) and arr.shape[-1] not in (1, 3):      # channels first (utils.load_image)
            arr = np.transpose(arr, (1, 2, 0))
        img = Image.fromarray(arr.astype(np.uint8))
    return img.convert("RGB") if img.mode != "RGB" else img


class Phi3VImageProcessor:
    model_input_names = ["pixel_values", "image_sizes"]

    def __init__(self, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD, num_crops: int = 4, num_img_tokens: int = 144,
                 **kwargs):
        self.image_mean, self.image_std = tuple(image_mean), tuple(image_std)
        self.num_crops, self.num_img_tokens, self.img_size = int(num_crops), int(num_img_tokens), TILE
        levels = np.arange(256, dtype=np.float32) / 255.0                                   # float32
        mean, std = np.array(self.image_mean), np.array(self.image_std)                      # float64
        self._lut = ((levels[None, :] - mean[:, None]) / std[:, None]).astype(np.float32)    # [3, 256]

    def calc_num_image_tokens(self, image) -> int:
        w, h = _to_pil(image).size
        hw, hh = calc_hd_transform_size(w, h, self.num_crops)
        nh, nw = hh // TILE, hw // TILE
        return (nh * nw + 1) * self.num_img_tokens + 1 + (nh + 1) * 12

    def _views(self, image) -> Tuple[np.ndarray, Tuple[int, int]]:
        """-> (float32 [1 + tiles, 3, 336, 336], (hd_height, hd_width))"""
        from PIL import Image

        img = _to_pil(image)
        tw, th = calc_hd_transform_size(img.size[0], img.size[1], self.num_crops)
        hd = img.resize((tw, th), Image.Resampling.BICUBIC)      # a multiple of 336 already: the reference's pad is a no-op
        glb = np.asarray(hd.resize((TILE, TILE), Image.Resampling.BICUBIC))
        full = np.asarray(hd)
        nh, nw = th // TILE, tw // TILE
        tiles = full.reshape(nh, TILE, nw, TILE, 3).transpose(0, 2, 1, 3, 4).reshape(nh * nw, TILE, TILE, 3)
        hwc = np.concatenate([glb[None], tiles], axis=0)
        out = np.empty((hwc.shape[0], 3, TILE, TILE), dtype=np.float32)
        for c in range(3):
            out[:, c] = self._lut[c].take(hwc[..., c])
        return out, (th, tw)

    def preprocess(self, images, return_tensors=None, **kwargs):
        if not isinstance(images, (list, tuple)):
            images = [images]
        views, sizes = zip(*(self._views(im) for im in images))
        T = max(v.shape[0] for v in views)
        pv = np.zeros((len(views), T, 3, TILE, TILE), dtype=np.float32)
        for i, v in enumerate(views):
            pv[i, : v.shape[0]] = v
        return {"pixel_values": pv, "image_sizes": np.array(sizes, dtype=np.int64)}

    __call__ = preprocess


_TAG = re.compile(r"<\|image_\d+\|>")


class Phi3VProcessor:
    """`processor(images=..., text=...)` -> input_ids int64 [1, L] (image positions negative), attention_mask,
    pixel_values float32 [B, T, 3, 336, 336], image_sizes int64 [B, 2]"""

    def __init__(self, image_processor=None, tokenizer=None, chat_template=None, **kwargs):
        self.image_processor = image_processor or Phi3VImageProcessor()
        self.tokenizer = tokenizer
        self.chat_template = chat_template

    def _convert(self, images: Sequence, text: str):
        pils = [_to_pil(im) for im in images]
        n_tok = [self.image_processor.calc_num_image_tokens(im) for im in pils]
        image_inputs = self.image_processor(pils) if pils else {}
        tags = _TAG.findall(text)
        if tags:
            tag_ids = [int(t.split("|")[1].split("_")[-1]) for t in tags]
            uniq = sorted(set(tag_ids))
            if uniq != list(range(1, len(uniq) + 1)):
                raise ValueError(f"Image IDs must be sequential starting from 1. Got: {uniq}")
            if len(uniq) != len(pils):
                raise ValueError(f"Number of image tags ({len(uniq)}) doesn't match number of images ({len(pils)})")
            runs = [[-i] * n_tok[i - 1] for i in tag_ids]
            bos = getattr(self.tokenizer, "bos_token_id", None)
            ids: List[int] = []
            for i, chunk in enumerate(_TAG.split(text)):
                toks = list(self.tokenizer.encode(chunk, add_special_tokens=(i == 0)))
                # the reference drops the FIRST token of every later chunk (as if it were a BOS), BOS or not
                # (processing_phi3_v.py:399-404: `offset = 1` for i > 0 in both branches)
                ids.extend(toks if i == 0 else toks[1:])
                if i < len(runs):
                    ids.extend(runs[i])
        else:
            ids = list(self.tokenizer.encode(text))
        out = {"input_ids": np.array([ids], dtype=np.int64), "attention_mask": np.ones((1, len(ids)), dtype=np.int32)}
        out.update(image_inputs)
        return out

    def __call__(self, images=None, text=None, **kwargs):
        if images is None and text is None:
            raise ValueError("You have to specify at least one of `images` or `text`.")
        images = [] if images is None else (list(images) if isinstance(images, (list, tuple)) else [images])
        if text is None:
            return self.image_processor(images) if images else {}
        texts = [text] if isinstance(text, str) else list(text)
        if len(texts) != 1:
            raise NotImplementedError("one prompt per call (batches go through batch_generate, one request each)")
        return self._convert(images, texts[0])

    def decode(self, *a, **k):
        return self.tokenizer.decode(*a, **k)

    def batch_decode(self, *a, **k):
        return self.tokenizer.batch_decode(*a, **k)

    def apply_chat_template(self, conversation, chat_template=None, add_generation_prompt=False, tokenize=False, **kwargs):
        chat_template = chat_template or self.chat_template or getattr(self.tokenizer, "chat_template", None)
        if chat_template is None:
            raise ValueError("No chat template found. Please provide a chat_template argument or ensure the tokenizer has one.")
        from jinja2 import Template

        rendered = Template(chat_template).render(messages=conversation, add_generation_prompt=add_generation_prompt, **kwargs)
        return self.tokenizer.encode(rendered) if tokenize else rendered
