"""LRU cache of projected image features for multi-turn conversations - the contract of the reference's
mlx_vlm/vision_cache.py:15-79 (`get` / `put` / `clear` / `len` / `in`, keys from paths, lists and image bytes).
Values are device tensors (the output of vision tower + projector, i.e. what `get_input_embeddings` accepts as
`cached_image_features`); a hit skips the whole ViT prefill of that turn."""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Any, Optional


class VisionFeatureCache:
    def __init__(self, max_size: int = 20):
        self.max_size = max_size
        self._cache: "OrderedDict[str, Any]" = OrderedDict()

    def _make_key(self, image_source: Any) -> str:
        """str / Path -> itself; list -> its members' keys joined; images / arrays -> a hash of their bytes"""
        if isinstance(image_source, str):
            return image_source
        if isinstance(image_source, (list, tuple)):
            return "|".join(self._make_key(x) for x in image_source)
        if hasattr(image_source, "tobytes"):
            shape = getattr(image_source, "shape", None) or getattr(image_source, "size", "")
            return "pil:" + hashlib.sha256(str(shape).encode() + image_source.tobytes()).hexdigest()[:16]
        return f"obj:{id(image_source)}"

    def get(self, image_source: Any) -> Optional[Any]:
        key = self._make_key(image_source)
        if key in self._cache:
            self._cache.move_to_end(key)
            return self._cache[key]
        return None

    def put(self, image_source: Any, features: Any) -> None:
        key = self._make_key(image_source)
        if key in self._cache:
            self._cache.move_to_end(key)
        elif len(self._cache) >= self.max_size:
            self._cache.popitem(last=False)
        self._cache[key] = features

    def clear(self) -> None:
        self._cache.clear()

    def __len__(self) -> int:
        return len(self._cache)

    def __contains__(self, image_source: Any) -> bool:
        return self._make_key(image_source) in self._cache
