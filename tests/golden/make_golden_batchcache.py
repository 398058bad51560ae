"""Golden vectors for the cache CONTRACT of SURVEY section 8(b) (round-4 review, item 4): `KVCache.update_and_fetch / state /
trim / extract` (reference models/cache.py:337-439) and `BatchKVCache.update_and_fetch / prepare / finalize / filter / extend /
extract / merge / trim / size / batch_size` (cache.py:972-1201), produced by the REFERENCE'S OWN classes executed over
oracle/mlx_shim (run once, in the build container):

    python tests/golden/make_golden_batchcache.py       # needs /root/reference; writes tests/golden/batchcache_ref.npz

A scenario is a list of operations on named caches; after every operation the bookkeeping of every live cache (offset,
left_padding, _idx, size, batch size) and the REAL part of its contents (row i: keys[i, :, left_padding[i]:_idx] - what the
reference's masks let attention see) are recorded.  tests/test_cache_contract_*.py replay the same operations on the paged
facades (mlx-vlm_amd/models/cache.py: block-table rows instead of tensor copies, no padding stored) and compare.  Keys / values
of token j of "stream" s are bf16 random vectors seeded by (s, j), so contents identify which token sits where."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden_ref as G  # noqa: E402  (puts the shim on sys.path)

H, D = 2, 128


def tok_kv(stream: int, j0: int, j1: int, pad: int = 0):
    """-> (k, v) float32 [H, pad + j1 - j0, D] of tokens j0..j1-1 of `stream`, `pad` zero rows in front (values exact in bf16)"""
    ks, vs = [], []
    for j in range(j0, j1):
        g = torch.Generator().manual_seed(100003 * stream + 17 * j + 1)
        ks.append((torch.randn(H, D, generator=g) * 0.5).to(torch.bfloat16).float())
        vs.append((torch.randn(H, D, generator=g) * 0.5).to(torch.bfloat16).float())
    k = torch.stack(ks, 1) if ks else torch.zeros(H, 0, D)
    v = torch.stack(vs, 1) if vs else torch.zeros(H, 0, D)
    z = torch.zeros(H, pad, D)
    return torch.cat([z, k], 1), torch.cat([z, v], 1)


# ---- the scenario: (op, args).  Streams are integers; a batch op names the stream of every row
SCENARIO = [
    ("new_kv", "a"), ("kv_update", "a", 1, 0, 5), ("kv_update", "a", 1, 5, 6), ("kv_update", "a", 1, 6, 7), ("kv_trim", "a", 2),
    ("kv_update", "a", 1, 5, 8),
    ("new_kv", "b"), ("kv_update", "b", 2, 0, 2),
    ("new_kv", "e"),
    # left-padded prompt batch: rows = streams 3, 4, 5 with 3, 1, 4 prompt tokens (padded to 4), then two decode steps
    ("new_batch", "B", [1, 3, 0]),
    ("batch_update", "B", [(3, 0, 3, 1), (4, 0, 1, 3), (5, 0, 4, 0)]),
    ("batch_update", "B", [(3, 3, 4, 0), (4, 1, 2, 0), (5, 4, 5, 0)]),
    ("batch_update", "B", [(3, 4, 5, 0), (4, 2, 3, 0), (5, 5, 6, 0)]),
    ("batch_extract", "B", 1, "x1"),
    ("batch_filter", "B", [0, 2]),
    ("batch_update", "B", [(3, 5, 6, 0), (5, 6, 7, 0)]),
    ("batch_filter", "B", [0]),                        # min left padding 1 -> the window shifts left
    ("batch_update", "B", [(3, 6, 7, 0)]),
    ("batch_trim", "B", 2),
    # merge of single caches (a: 8 tokens, b: 2, e: empty) and a decode step on the merged batch
    ("merge", "M", ["a", "b", "e"]),
    ("batch_update", "M", [(1, 8, 9, 0), (2, 2, 3, 0), (6, 0, 1, 0)]),
    ("batch_extract", "M", 1, "x2"),
    # a second batch joins (extend): two rows with 2 and 3 tokens
    ("new_batch", "N", [1, 0]),
    ("batch_update", "N", [(7, 0, 2, 1), (8, 0, 3, 0)]),
    ("batch_extend", "M", "N"),
    ("batch_update", "M", [(1, 9, 10, 0), (2, 3, 4, 0), (6, 1, 2, 0), (7, 2, 3, 0), (8, 3, 4, 0)]),
    ("batch_filter", "M", [1, 3, 4]),
    # right-padded prefill chunk + finalize (the reference's batched chunked prefill, cache.py:1027-1048)
    ("new_batch", "R", [0, 0, 0]),
    ("batch_prepare", "R", dict(right_padding=[0, 2, 1])),
    ("batch_update_right", "R", [(9, 0, 4), (10, 0, 2), (11, 0, 3)], 4),
    ("batch_finalize", "R"),
    ("batch_update", "R", [(9, 4, 5, 0), (10, 2, 3, 0), (11, 3, 4, 0)]),
    # an empty batch cache extended by another empty one, then prepared with left padding
    ("new_batch", "E1", [0]), ("new_batch", "E2", [0, 0]), ("batch_extend", "E1", "E2"),
    ("batch_prepare", "E1", dict(left_padding=[2, 0, 1])),
    ("batch_update", "E1", [(12, 0, 1, 2), (13, 0, 3, 0), (14, 0, 2, 1)]),
]


def main():
    mx, q, cfgm, cache_mod, su = G.import_reference()
    BF = torch.bfloat16
    arr = lambda t: mx.array(t.to(BF))           # noqa: E731  (bf16 arrays as the model's caches hold)
    live = {}
    out = {}
    n_kv = lambda c: 0 if c.keys is None else int(c.offset)       # noqa: E731

    def snap(step):
        for name, c in live.items():
            p = f"s{step:02d}.{name}."
            if isinstance(c, cache_mod.BatchKVCache):
                lp = np.asarray(c.left_padding._t.numpy()).astype(np.int64)
                off = np.asarray(c.offset._t.numpy()).astype(np.int64)
                out[p + "left_padding"], out[p + "offset"] = lp, off
                out[p + "idx"] = np.array(int(c._idx)); out[p + "size"] = np.array(int(c.size()))
                out[p + "batch_size"] = np.array(int(c.batch_size)); out[p + "empty"] = np.array(bool(c.empty()))
                out[p + "nbytes"] = np.array(int(c.nbytes))
                if c.keys is not None:
                    for i in range(len(lp)):
                        out[p + f"k{i}"] = c.keys._t[i, :, int(lp[i]):int(c._idx)].float().numpy()
                        out[p + f"v{i}"] = c.values._t[i, :, int(lp[i]):int(c._idx)].float().numpy()
            else:
                out[p + "offset"] = np.array(int(c.offset)); out[p + "size"] = np.array(int(c.size()))
                out[p + "empty"] = np.array(bool(c.empty())); out[p + "nbytes"] = np.array(int(c.nbytes))
                if c.keys is not None:
                    k, v = c.state
                    out[p + "k0"] = k._t[0].float().numpy(); out[p + "v0"] = v._t[0].float().numpy()

    for step, op in enumerate(SCENARIO):
        kind = op[0]
        if kind == "new_kv":
            live[op[1]] = cache_mod.KVCache()
        elif kind == "kv_update":
            k, v = tok_kv(op[2], op[3], op[4])
            rk, rv = live[op[1]].update_and_fetch(arr(k[None]), arr(v[None]))
            assert rk.shape[2] == live[op[1]].offset
        elif kind == "kv_trim":
            out[f"s{step:02d}.ret"] = np.array(int(live[op[1]].trim(op[2])))
        elif kind == "new_batch":
            live[op[1]] = cache_mod.BatchKVCache(list(op[2]))
        elif kind == "batch_update":
            S = max(j1 - j0 + pad for _, j0, j1, pad in op[2])
            ks, vs = zip(*[tok_kv(s, j0, j1, pad) for s, j0, j1, pad in op[2]])
            assert all(k.shape[1] == S for k in ks)
            live[op[1]].update_and_fetch(arr(torch.stack(ks)), arr(torch.stack(vs)))
        elif kind == "batch_update_right":
            S = op[3]
            ks, vs = [], []
            for s, j0, j1 in op[2]:
                k, v = tok_kv(s, j0, j1)
                z = torch.zeros(H, S - (j1 - j0), D)
                ks.append(torch.cat([k, z], 1)); vs.append(torch.cat([v, z], 1))
            live[op[1]].update_and_fetch(arr(torch.stack(ks)), arr(torch.stack(vs)))
        elif kind == "batch_prepare":
            live[op[1]].prepare(**op[2])
        elif kind == "batch_finalize":
            live[op[1]].finalize()
        elif kind == "batch_filter":
            live[op[1]].filter(mx.array(np.asarray(op[2], dtype=np.int32)))
        elif kind == "batch_extend":
            live[op[1]].extend(live.pop(op[2]))
        elif kind == "batch_extract":
            live[op[3]] = live[op[1]].extract(op[2])
        elif kind == "batch_trim":
            out[f"s{step:02d}.ret"] = np.array(int(live[op[1]].trim(op[2])))
        elif kind == "merge":
            live[op[1]] = cache_mod.BatchKVCache.merge([live[n] for n in op[2]])
        else:
            raise ValueError(kind)
        snap(step)
    out["scenario_json"] = np.array(json.dumps(SCENARIO))
    dst = os.path.join(HERE, "batchcache_ref.npz")
    np.savez_compressed(dst, **out)
    print(f"wrote {dst}: {len(out)} arrays, {os.path.getsize(dst) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
