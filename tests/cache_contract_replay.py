"""Replays the scenario of tests/golden/make_golden_batchcache.py on the paged facades (mlx-vlm_amd/models/cache.py) and compares
every recorded number with what the reference's own KVCache / BatchKVCache gave (batchcache_ref.npz).  Every operation is applied
to the facade of EVERY layer, as a model's cache list is used.  check_contents=False: bookkeeping only (the CPU run, with the
device write stubbed out)."""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
H, D = 2, 128


def tok_kv(stream, j0, j1, pad=0):      # the generator script's token vectors (tests/golden/make_golden_batchcache.py)
    ks, vs = [], []
    for j in range(j0, j1):
        g = torch.Generator().manual_seed(100003 * stream + 17 * j + 1)
        ks.append((torch.randn(H, D, generator=g) * 0.5).to(torch.bfloat16).float())
        vs.append((torch.randn(H, D, generator=g) * 0.5).to(torch.bfloat16).float())
    k = torch.stack(ks, 1) if ks else torch.zeros(H, 0, D)
    v = torch.stack(vs, 1) if vs else torch.zeros(H, 0, D)
    z = torch.zeros(H, pad, D)
    return torch.cat([z, k], 1), torch.cat([z, v], 1)


def replay(pool, device, check_contents=True):
    from mlx_vlm_amd.models.cache import BatchKVCache, KVCache, PagedSequence

    z = np.load(os.path.join(HERE, "golden", "batchcache_ref.npz"))
    scenario = json.loads(str(z["scenario_json"]))
    L = pool.n_layers
    live = {}          # name -> list of per-layer facades
    alias_of = {}      # extracted name -> the batch it shares a row with
    BF = torch.bfloat16
    dev = lambda t: t.to(BF).to(device)      # noqa: E731
    n_checked = 0

    def check(step):
        nonlocal n_checked
        for name, layers in live.items():
            p = f"s{step:02d}.{name}."
            for l, c in enumerate(layers):
                if isinstance(c, BatchKVCache):
                    assert c.left_padding.tolist() == z[p + "left_padding"].tolist(), (step, name, l, "left_padding", c.left_padding, z[p + "left_padding"])
                    assert c.offset.tolist() == z[p + "offset"].tolist(), (step, name, l, "offset", c.offset, z[p + "offset"])
                    assert c._idx == int(z[p + "idx"]) and c.size() == int(z[p + "size"]), (step, name, l, c._idx)
                    assert c.batch_size == int(z[p + "batch_size"]) and c.empty() == bool(z[p + "empty"]), (step, name, l)
                    assert (c.nbytes > 0) == (int(z[p + "nbytes"]) > 0) or not check_contents, (step, name, l, c.nbytes)
                    for i in range(c.batch_size):       # a row holds exactly the tokens the reference's window shows for it
                        want = max(0, int(z[p + "offset"][i])) if (p + f"k{i}") in z.files else 0
                        assert c._kept(i) == want, (step, name, l, i, c._kept(i), want)
                    if check_contents and (p + "k0") in z.files:
                        k, v, _, _ = c.state
                        for i in range(c.batch_size):
                            lp = int(c.left_padding[i])
                            assert torch.equal(k[i, :, lp:c._idx].float().cpu(), torch.from_numpy(z[p + f"k{i}"])), (step, name, l, i, "k")
                            assert torch.equal(v[i, :, lp:c._idx].float().cpu(), torch.from_numpy(z[p + f"v{i}"])), (step, name, l, i, "v")
                            assert float(k[i, :, :lp].abs().sum()) == 0.0          # padding: zeros here (masked in the reference)
                            n_checked += 1
                else:
                    assert c.offset == int(z[p + "offset"]) and c.size() == int(z[p + "size"]), (step, name, l, c.offset, int(z[p + "offset"]))
                    assert c.empty() == bool(z[p + "empty"]), (step, name, l)
                    if check_contents and (p + "k0") in z.files:
                        k, v = c.state
                        assert torch.equal(k[0].float().cpu(), torch.from_numpy(z[p + "k0"])), (step, name, l, "k")
                        assert torch.equal(v[0].float().cpu(), torch.from_numpy(z[p + "v0"])), (step, name, l, "v")
                        n_checked += 1

    for step, op in enumerate(scenario):
        kind = op[0]
        if kind == "new_kv":
            seq = PagedSequence(pool)
            live[op[1]] = [KVCache(seq, l) for l in range(L)]
        elif kind == "kv_update":
            k, v = tok_kv(op[2], op[3], op[4])
            before = live[op[1]][0].offset
            for l, c in enumerate(live[op[1]]):
                rk, rv = c.update_and_fetch(dev(k[None]), dev(v[None]))
                assert rk.shape[2] == c.offset == before + (op[4] - op[3])
                if l + 1 < L:      # the layers behind have not seen the tokens yet; the engine's counter follows the slowest
                    assert live[op[1]][l + 1].offset == before and c._seq.offset == before
            assert live[op[1]][0]._seq.offset == before + (op[4] - op[3]) and live[op[1]][0]._seq._layer_off is None
        elif kind == "kv_trim":
            rets = [c.trim(op[2]) for c in live[op[1]]]
            assert rets[0] == int(z[f"s{step:02d}.ret"])
        elif kind == "new_batch":
            live[op[1]] = BatchKVCache.for_layers(pool, list(op[2]))
        elif kind in ("batch_update", "batch_update_right"):
            if kind == "batch_update":
                ks, vs = zip(*[tok_kv(s, j0, j1, pad) for s, j0, j1, pad in op[2]])
            else:
                S = op[3]
                ks, vs = [], []
                for s, j0, j1 in op[2]:
                    k, v = tok_kv(s, j0, j1)
                    zz = torch.zeros(H, S - (j1 - j0), D)
                    ks.append(torch.cat([k, zz], 1)); vs.append(torch.cat([v, zz], 1))
            K, V = dev(torch.stack(list(ks))), dev(torch.stack(list(vs)))
            # an extracted cache SHARES its row's pages (the reference copies the row): once the batch appends to that row the
            # two diverge by construction - such a cache is compared up to here and not beyond
            for name in [n for n, src in alias_of.items() if src == op[1] and any(r is live[n][0]._seq for r in live[op[1]][0]._rows)]:
                live.pop(name); alias_of.pop(name)
            for c in live[op[1]]:
                rk, rv = c.update_and_fetch(K, V)
                assert rk.shape[2] == c._idx
        elif kind == "batch_prepare":
            for c in live[op[1]]:
                c.prepare(**op[2])
        elif kind == "batch_finalize":
            for c in live[op[1]]:
                c.finalize()
        elif kind == "batch_filter":
            for c in live[op[1]]:
                c.filter(np.asarray(op[2], dtype=np.int32))
        elif kind == "batch_extend":
            other = live.pop(op[2])
            for c, o in zip(live[op[1]], other):
                c.extend(o)
        elif kind == "batch_extract":
            live[op[3]] = [c.extract(op[2]) for c in live[op[1]]]
            alias_of[op[3]] = op[1]
        elif kind == "batch_trim":
            rets = [c.trim(op[2]) for c in live[op[1]]]
            assert rets[0] == int(z[f"s{step:02d}.ret"])
        elif kind == "merge":
            # merge MOVES the single caches' rows into the batch (the reference copies them into a padded tensor; its callers
            # drop the singles, ar.py:743-746): the singles are not compared any further
            live[op[1]] = [BatchKVCache.merge([live[n][l] for n in op[2]]) for l in range(L)]
            for n in op[2]:
                live.pop(n)
        else:
            raise ValueError(kind)
        check(step)
    return n_checked, len(scenario)
