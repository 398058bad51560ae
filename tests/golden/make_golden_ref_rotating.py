"""Golden vectors for `max_kv_size` (SURVEY section 2 listed it out of scope; round-3 review: the last refused option of
generate_step), produced by the REFERENCE'S OWN files executed over oracle/mlx_shim (run once, in the build container):

    python tests/golden/make_golden_ref_rotating.py      # needs /root/reference; writes tests/golden/rotating_ref.npz

    models/cache.py:45-70       make_prompt_cache(model, max_kv_size): RotatingKVCache(max_size, keep=4) per layer (none of the
                                built families defines make_cache)
    models/cache.py:442-625     RotatingKVCache: _update_concat (the prompt, kept whole), _update_in_place (decode: grow to
                                max_size in 256-steps, trim a longer prompt to keep + the most recent, then overwrite the oldest
                                non-sink entry in ring order), make_mask
    models/base.py:214-228      create_attention_mask(h, cache): `cache` is the LIST of per-layer caches, so the prompt is
                                attended with the plain causal mask whatever its length
    generate/ar.py:151-515      generate_step(max_kv_size=...)

Recorded: a teacher-forced decode (every step's logits, and after every step which TOKEN INDICES the layer-0 cache holds, read
back from a tagged copy of the cache's own bookkeeping) for a prompt shorter than max_kv_size (the ring fills during the
decode), one exactly max_kv_size long, and one longer (trimmed at the first decode step); generate_step tokens + log-probs on
the peaked head for the short and the long prompt; RotatingKVCache.trim / is_trimmable / size known answers."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden_ref as G  # noqa: E402  (puts the shim on sys.path)
from make_golden_ref_kvquant import build_model  # noqa: E402

BF = torch.bfloat16


def held_tokens(mx, cache_mod, max_size, keep, n_prompt, n_steps):
    """Which token indices a RotatingKVCache holds after the prompt and after every decode step: the reference's class run on
    keys whose first element is the token index."""
    c = cache_mod.RotatingKVCache(max_size=max_size, keep=keep)
    tag = lambda a, b: mx.array(np.arange(a, b, dtype=np.float32).reshape(1, 1, b - a, 1))       # noqa: E731
    k, _ = c.update_and_fetch(tag(0, n_prompt), tag(0, n_prompt))
    out = [np.sort(np.asarray(k._t.numpy()).reshape(-1).astype(np.int64))]
    for t in range(n_prompt, n_prompt + n_steps):
        k, _ = c.update_and_fetch(tag(t, t + 1), tag(t, t + 1))
        out.append(np.sort(np.asarray(k._t.numpy()).reshape(-1).astype(np.int64)))
    return out, c


def fresh(model):
    """a model object keeps the rope positions of its last prompt (language.py:404-470); every run here starts like a new one"""
    model.language_model._position_ids = None
    model.language_model._rope_deltas = None


def main():
    from oracle import qwen2_vl as oq

    torch.manual_seed(0)
    torch.set_num_threads(4)
    mx, q, cfgm, cache_mod, su = G.import_reference()
    ar = q._generate_ar
    for m in (ar, cache_mod):
        assert m.__file__.startswith(G.REF), m.__file__
    blob = {}
    MAXS, KEEP = 24, 4

    # ---------------------------------------------------------------- the class itself: which tokens are held, known answers
    for name, n_prompt in (("short", 9), ("exact", 24), ("long", 41)):
        sets, c = held_tokens(mx, cache_mod, MAXS, KEEP, n_prompt, 40)
        width = max(len(s) for s in sets)
        blob[f"held.{name}"] = np.stack([np.pad(s, (0, width - len(s)), constant_values=-1) for s in sets])
        blob[f"held.{name}.n_prompt"] = np.array(n_prompt)
        blob[f"held.{name}.size_offset"] = np.array([c.size(), c.offset])
    c = cache_mod.RotatingKVCache(max_size=MAXS, keep=KEEP)
    z = lambda n: mx.zeros((1, 1, n, 2))                                                            # noqa: E731
    c.update_and_fetch(z(10), z(10))
    ka = [int(c.is_trimmable()), c.trim(3), c.offset, c.size()]
    for _ in range(20):
        c.update_and_fetch(z(1), z(1))
    ka += [int(c.is_trimmable()), c.offset, c.size()]
    blob["class.known_answers"] = np.array(ka, dtype=np.int64)

    # ---------------------------------------------------------------- teacher-forced decode through the model
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    model = build_model(mx, q, cfgm, cfg, W)
    rng = np.random.default_rng(51)
    forced = rng.integers(3, 1000, 30)
    blob["tf.forced"] = forced.astype(np.int64)
    for name, n_prompt in (("short", 9), ("exact", 24), ("long", 41)):
        ids = rng.integers(3, 1000, (1, n_prompt)).astype(np.int32)
        blob[f"tf.{name}.input_ids"] = ids.astype(np.int64)
        kv = cache_mod.make_prompt_cache(model.language_model, max_kv_size=MAXS)
        assert isinstance(kv[0], cache_mod.RotatingKVCache) and kv[0].keep == KEEP
        fresh(model)
        emb = model.get_input_embeddings(mx.array(ids), None)
        out = model.language_model(mx.array(ids), inputs_embeds=emb.inputs_embeds, cache=kv)
        rows = [G.f32(out.logits[0, -1])]
        for y in forced:
            o = model.language_model(mx.array(np.array([[int(y)]], dtype=np.int32)), cache=kv)
            rows.append(G.f32(o.logits[0, -1]))
        blob[f"tf.{name}.logits"] = np.stack(rows)
        # the same decode WITHOUT a bound: the two must part once the ring starts to drop tokens
        kv2 = [cache_mod.KVCache() for _ in model.language_model.layers]
        fresh(model)
        out = model.language_model(mx.array(ids), inputs_embeds=emb.inputs_embeds, cache=kv2)
        rows2 = [G.f32(out.logits[0, -1])]
        for y in forced:
            rows2.append(G.f32(model.language_model(mx.array(np.array([[int(y)]], dtype=np.int32)), cache=kv2).logits[0, -1]))
        same = [bool(np.array_equal(a, b)) for a, b in zip(rows, rows2)]
        blob[f"tf.{name}.equals_unbounded"] = np.array(same)
        print(f"teacher-forced {name}: rows equal to the unbounded cache: {same.index(False) if False in same else len(same)} of {len(same)}")

    # ---------------------------------------------------------------- generate_step(max_kv_size=...) on the peaked head
    cfgp = oq.tiny_cfg()
    cfgp.text.tie_word_embeddings = False
    Wp = oq.random_weights(cfgp, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    for k in list(Wp):
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            Wp[k] = (Wp[k].float() * 0.5).to(BF)
    Wp = oq.peak_head(Wp, cfgp, gamma=4.0, stride=389, n_cycle=1000)
    pmodel = build_model(mx, q, cfgm, cfgp, Wp)
    for name, n_prompt in (("short", 11), ("long", 37)):
        ids = np.random.default_rng(60 + n_prompt).integers(3, 1000, (1, n_prompt)).astype(np.int32)
        toks, lps = [], []
        fresh(pmodel)
        for tok, lp in ar.generate_step(mx.array(ids), pmodel, None, None, max_tokens=28, temperature=0.0, max_kv_size=MAXS):
            toks.append(int(tok))
            lps.append(G.f32(lp))
        blob[f"gen.{name}.input_ids"] = ids.astype(np.int64)
        blob[f"gen.{name}.tokens"], blob[f"gen.{name}.logprobs"] = np.array(toks, dtype=np.int64), np.stack(lps)
        print("generate_step", name, toks)
    blob["max_kv_size"], blob["keep"] = np.array(MAXS), np.array(KEEP)
    np.savez_compressed(os.path.join(HERE, "rotating_ref.npz"), **blob)
    print("wrote rotating_ref.npz:", len(blob), "arrays")


if __name__ == "__main__":
    main()
