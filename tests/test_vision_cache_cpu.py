"""VisionFeatureCache (mlx-vlm_amd/vision_cache.py): the reference's contract (mlx_vlm/vision_cache.py:15-79 - least recently used
entry goes first, `get` refreshes, `put` of a present key replaces and refreshes, keys from paths / lists / image bytes) plus the
byte accounting this engine adds for device tensors."""
import numpy as np
import torch

from mlx_vlm_amd.vision_cache import VisionFeatureCache, source_key


def test_lru_order_get_refreshes_and_put_replaces():
    c = VisionFeatureCache(max_size=3)
    for k in "abc":
        c.put(k, torch.zeros(2))
    assert len(c) == 3 and "a" in c
    assert c.get("a") is not None                  # a is now the most recent
    c.put("d", torch.zeros(2))                     # evicts b, the least recent
    assert "b" not in c and all(k in c for k in "acd")
    c.put("c", torch.ones(2))                      # replace + refresh: c newest, a oldest
    assert torch.equal(c.get("c"), torch.ones(2))
    c.put("e", torch.zeros(2))
    assert "a" not in c and "d" in c and "c" in c and "e" in c
    assert c.get("nope") is None
    s = c.stats()
    assert s["entries"] == 3 and s["evictions"] == 2 and s["misses"] == 1 and s["hits"] == 2
    c.clear()
    assert len(c) == 0 and c.nbytes == 0


def test_keys_paths_lists_and_image_bytes():
    a = np.arange(12, dtype=np.uint8).reshape(2, 2, 3)
    b = a.copy()
    assert source_key("cat.png") == "cat.png"
    assert source_key(["x.png", "y.png"]) == "x.png|y.png" != source_key(["y.png", "x.png"])
    assert source_key(a) == source_key(b) and source_key(a).startswith("pil:")
    assert source_key(a) != source_key(a.reshape(4, 3))          # same bytes, other shape: another image
    b[0, 0, 0] = 99
    assert source_key(a) != source_key(b)
    c = VisionFeatureCache()
    c.put([a, "z.png"], torch.zeros(1))
    assert [a.copy(), "z.png"] in c and c._make_key("q") == "q"


def test_byte_budget_counts_tensors_and_lists_of_tensors():
    c = VisionFeatureCache(max_size=10, max_bytes=4096)
    c.put("a", torch.zeros(256, dtype=torch.float32))                               # 1024 B
    c.put("b", [torch.zeros(256, dtype=torch.bfloat16), torch.zeros(256, dtype=torch.bfloat16)])      # 1024 B
    assert c.nbytes == 2048
    c.put("c", torch.zeros(640, dtype=torch.float32))                               # 2560 B: a has to go
    assert "a" not in c and "b" in c and c.nbytes == 1024 + 2560
    c.put("huge", torch.zeros(4096, dtype=torch.float32))                           # over the whole budget: kept alone
    assert len(c) == 1 and "huge" in c and c.nbytes == 16384
    c.put("huge", torch.zeros(1))                                                   # replaced: the accounting follows
    assert c.nbytes == 4
