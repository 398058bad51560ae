"""Image processor and prompt expansion of Idefics2 - host mirror of what the reference runs for this model: transformers'
`Idefics2ImageProcessor` (resize to shortest_edge 378 / longest_edge 980 with PIL bilinear, x * (1 / 255), (x - mean) / std,
zero padding to the largest image of the call + `pixel_attention_mask`, all-zero padding images for samples with fewer
images, optional 4 + 1 image splitting) wrapped by `mlx_vlm/models/idefics2/processing_idefics2.py:33-217` (every `<image>`
becomes `<fake_token_around_image>` + image_seq_len x `<image>` + `<fake_token_around_image>`, five times when splitting;
adjacent fake tokens are merged).  Bit-exact against transformers' PIL backend (tests/test_oracle_ref_golden_idefics2.py,
tests/test_idefics2_cpu.py)."""
from __future__ import annotations

import re
from typing import List, Optional, Sequence

import numpy as np


def resize_output_size(height: int, width: int, shortest_edge: int, longest_edge: int):
    aspect = width / height
    if width >= height and width > longest_edge:
        width = longest_edge
        height = int(width / aspect)
    elif height > width and height > longest_edge:
        height = longest_edge
        width = int(height * aspect)
    return max(height, shortest_edge), max(width, shortest_edge)


def _hwc_u8(img) -> np.ndarray:
    from PIL import Image

    if isinstance(img, str):
        img = Image.open(img)
    if isinstance(img, Image.Image):
        return np.asarray(img.convert("RGB"))
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[0] in (1, 3) and a.shape[-1] not in (1, 3):      # channels first (utils.load_image)
        a = np.transpose(a, (1, 2, 0))
    if a.ndim == 2:
        a = np.stack([a] * 3, axis=-1)
    return a[..., :3].astype(np.uint8)


class Idefics2ImageProcessor:
    model_input_names = ["pixel_values", "pixel_attention_mask"]

    def __init__(self, size=None, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), rescale_factor=1 / 255,
                 do_image_splitting: bool = False, **kwargs):
        size = size or {"shortest_edge": 378, "longest_edge": 980}
        self.size = dict(size)
        self.image_mean, self.image_std, self.rescale_factor = tuple(image_mean), tuple(image_std), rescale_factor
        self.do_image_splitting = bool(do_image_splitting)
        # rescale (float64 product cast to float32) and normalise (float32) are functions of (channel, byte): one table
        levels = (np.arange(256, dtype=np.float64) * self.rescale_factor).astype(np.float32)
        mean, std = np.asarray(self.image_mean, dtype=np.float32), np.asarray(self.image_std, dtype=np.float32)
        self._lut = (levels[None, :] - mean[:, None]) / std[:, None]

    def _one(self, a: np.ndarray) -> List[np.ndarray]:
        from PIL import Image

        parts = [a]
        if self.do_image_splitting:
            mh, mw = a.shape[0] // 2, a.shape[1] // 2
            parts = [a[:mh, :mw], a[:mh, mw:], a[mh:, :mw], a[mh:, mw:], a]
        out = []
        for part in parts:
            h, w = resize_output_size(part.shape[0], part.shape[1], self.size["shortest_edge"], self.size["longest_edge"])
            hwc = np.asarray(Image.fromarray(np.ascontiguousarray(part)).resize((w, h), resample=Image.BILINEAR))
            out.append(np.stack([self._lut[c].take(hwc[:, :, c]) for c in range(3)]))
        return out

    def preprocess(self, images, return_tensors=None, **kwargs):
        """images: one image, a list of images (one sample) or a list of lists (samples).  -> pixel_values float32
        [B, N, 3, H, W], pixel_attention_mask int64 [B, N, H, W]"""
        if not isinstance(images, (list, tuple)):
            images = [[images]]
        elif images and not isinstance(images[0], (list, tuple)):
            images = [list(images)]
        rows = [[x for im in sample for x in self._one(_hwc_u8(im))] for sample in images]
        N = max(len(r) for r in rows)
        H = max(x.shape[1] for r in rows for x in r)
        W = max(x.shape[2] for r in rows for x in r)
        pv = np.zeros((len(rows), N, 3, H, W), dtype=np.float32)
        mask = np.zeros((len(rows), N, H, W), dtype=np.int64)
        for i, r in enumerate(rows):
            for j, x in enumerate(r):
                pv[i, j, :, : x.shape[1], : x.shape[2]] = x
                mask[i, j, : x.shape[1], : x.shape[2]] = 1
        return {"pixel_values": pv, "pixel_attention_mask": mask}

    __call__ = preprocess


class Idefics2Processor:
    """`processor(images=..., text=...)` -> input_ids / attention_mask (numpy, from the HF tokenizer) + pixel_values /
    pixel_attention_mask"""

    def __init__(self, image_processor=None, tokenizer=None, image_seq_len: int = 64, chat_template: Optional[str] = None, **kwargs):
        self.image_processor = image_processor or Idefics2ImageProcessor()
        self.tokenizer = tokenizer
        self.image_seq_len = int(image_seq_len)
        self.chat_template = chat_template
        self.fake_image_token = getattr(tokenizer, "image_boundary_token", None) or "<fake_token_around_image>"
        self.image_token = getattr(tokenizer, "image_token", None) or "<image>"
        self.image_token_id = tokenizer.convert_tokens_to_ids(self.image_token) if hasattr(tokenizer, "convert_tokens_to_ids") else None

    def expand_prompt(self, text: str) -> str:
        fake, image = self.fake_image_token, self.image_token
        image_str = f"{fake}{image * self.image_seq_len}{fake}"
        if self.image_processor.do_image_splitting:
            image_str = image_str * 5
        s = text.replace(image, image_str).replace(f"{fake}{fake}", fake)
        return re.sub(rf"{re.escape(fake)}(?=[^\s<])", f"{fake} ", s)

    def __call__(self, images=None, text=None, **kwargs):
        if text is None and images is None:
            raise ValueError("You must provide either `text` or `images`.")
        kwargs.pop("return_tensors", None)
        out = {}
        n_in_text: List[int] = []
        if text is not None:
            texts = [text] if isinstance(text, str) else list(text)
            n_in_text = [t.count(self.image_token) for t in texts]
            enc = self.tokenizer([self.expand_prompt(t) for t in texts], **kwargs)
            out["input_ids"] = np.asarray(enc["input_ids"], dtype=np.int64)
            out["attention_mask"] = np.asarray(enc["attention_mask"], dtype=np.int32)
        if images is not None and (not isinstance(images, (list, tuple)) or len(images)):
            if not isinstance(images, (list, tuple)):
                images = [[images]]
            elif not isinstance(images[0], (list, tuple)):
                images = list(images)
                if text is not None:
                    if sum(n_in_text) != len(images):
                        raise ValueError(f"The total number of {self.image_token} tokens in the prompts should be the same as the "
                                         f"number of images passed. Found {sum(n_in_text)} {self.image_token} tokens and "
                                         f"{len(images)} images.")
                    cuts = np.cumsum([0] + n_in_text)
                    images = [images[cuts[i]:cuts[i + 1]] for i in range(len(n_in_text))]
                else:
                    images = [images]
            if text is not None and [len(s) for s in images] != n_in_text:
                raise ValueError(f"The number of images in the text {n_in_text} and images {[len(s) for s in images]} should be the same.")
            out.update(self.image_processor(images))
        return out

    def decode(self, *a, **k):
        return self.tokenizer.decode(*a, **k)

    def batch_decode(self, *a, **k):
        return self.tokenizer.batch_decode(*a, **k)
