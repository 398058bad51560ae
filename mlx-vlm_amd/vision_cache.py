"""Projected image features kept across the turns of a conversation: the contract of the reference's `VisionFeatureCache`
(mlx_vlm/vision_cache.py:15-79 - `get` / `put` / `clear` / `len` / `in`, at most `max_size` entries, least recently used goes
first; keys from paths, lists of sources and image bytes).  A hit skips the whole ViT prefill of that turn: the values are what
`get_input_embeddings` accepts as `cached_image_features` (the output of vision tower + projector).

Here the values are DEVICE tensors, so the cache also knows what it holds in HBM: every entry is recorded with its byte count and
a use tick, `max_bytes` (optional) bounds the total next to the entry count, and `stats()` reports hits / misses / evictions /
bytes for the serving loop's metrics.  Recency is the tick, eviction takes the entry with the smallest one."""
from __future__ import annotations

import hashlib
from typing import Any, Dict, Optional


def _nbytes(value: Any) -> int:
    """bytes a cached value holds: a tensor, or a list / tuple / dict of tensors (per-image feature lists)"""
    if hasattr(value, "element_size") and hasattr(value, "numel"):
        return int(value.element_size()) * int(value.numel())
    if isinstance(value, dict):
        return sum(_nbytes(v) for v in value.values())
    if isinstance(value, (list, tuple)):
        return sum(_nbytes(v) for v in value)
    return 0


def source_key(image_source: Any) -> str:
    """One string per image source: a path / URL is its own key, a sequence joins its members' keys (order matters: it is the
    order of the image tokens), anything that can give its bytes (PIL image, ndarray, tensor on the host) is a digest of shape +
    bytes, any other object is keyed by identity."""
    if isinstance(image_source, (str, bytes)):
        return image_source if isinstance(image_source, str) else "bytes:" + hashlib.sha256(image_source).hexdigest()[:16]
    if hasattr(image_source, "__fspath__"):
        return str(image_source.__fspath__())
    if isinstance(image_source, (list, tuple)):
        return "|".join(source_key(x) for x in image_source)
    if hasattr(image_source, "tobytes"):
        dims = getattr(image_source, "shape", None) or getattr(image_source, "size", "")
        h = hashlib.sha256(repr(tuple(dims) if hasattr(dims, "__iter__") else dims).encode())
        h.update(image_source.tobytes())
        return "pil:" + h.hexdigest()[:16]
    return f"obj:{id(image_source)}"


class VisionFeatureCache:
    def __init__(self, max_size: int = 20, max_bytes: Optional[int] = None):
        self.max_size = int(max_size)
        self.max_bytes = None if max_bytes is None else int(max_bytes)
        self._entries: Dict[str, list] = {}            # key -> [value, nbytes, tick of the last use]
        self._tick = 0
        self._bytes = 0
        self._hits = self._misses = self._evictions = 0

    # the reference's private name for the key function (vision_cache.py:22-42); kept for code that calls it
    def _make_key(self, image_source: Any) -> str:
        return source_key(image_source)

    def _touch(self, entry: list) -> None:
        self._tick += 1
        entry[2] = self._tick

    def _evict_one(self) -> None:
        oldest = min(self._entries, key=lambda k: self._entries[k][2])
        self._bytes -= self._entries.pop(oldest)[1]
        self._evictions += 1

    def get(self, image_source: Any) -> Optional[Any]:
        entry = self._entries.get(source_key(image_source))
        if entry is None:
            self._misses += 1
            return None
        self._hits += 1
        self._touch(entry)
        return entry[0]

    def put(self, image_source: Any, features: Any) -> None:
        key, size = source_key(image_source), _nbytes(features)
        old = self._entries.pop(key, None)
        if old is not None:
            self._bytes -= old[1]
        # make room: one slot for the newcomer, and (when a byte budget is set) its bytes - a value larger than the whole budget
        # is still kept, alone (a turn must be able to reuse what it just computed)
        while self._entries and (len(self._entries) >= self.max_size
                                 or (self.max_bytes is not None and self._bytes + size > self.max_bytes)):
            self._evict_one()
        entry = [features, size, 0]
        self._touch(entry)
        self._entries[key] = entry
        self._bytes += size

    def clear(self) -> None:
        self._entries.clear()
        self._bytes = 0

    def stats(self) -> dict:
        return {"entries": len(self._entries), "bytes": self._bytes, "hits": self._hits, "misses": self._misses,
                "evictions": self._evictions}

    @property
    def nbytes(self) -> int:
        return self._bytes

    def __len__(self) -> int:
        return len(self._entries)

    def __contains__(self, image_source: Any) -> bool:
        return source_key(image_source) in self._entries
