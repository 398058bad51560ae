"""`max_kv_size` (reference RotatingKVCache, models/cache.py:442-625, built by make_prompt_cache for the families of this
package): the oracle's restatement (oracle/ops.py::RotatingKVCache, oracle/qwen2_vl.py max_kv_size=) against the reference's
own files executed over the shim (tests/golden/make_golden_ref_rotating.py -> rotating_ref.npz) - which tokens the cache holds
after every step, every logit row of teacher-forced decodes whose ring fills / starts full / starts over-full, generate_step's
tokens and bf16 log-probs, the class's trim / size known answers.  Bit for bit (rope_mode "fallback": the reference's pure-MLX
rotation, which is what runs off-Metal - tests/test_oracle_ref_golden.py states the fused kernel's own contract)."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import qwen2_vl as oq

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotating_ref.npz"))
BF = torch.bfloat16
MAXS, KEEP = int(G["max_kv_size"]), int(G["keep"])


@pytest.mark.parametrize("name", ["short", "exact", "long"])
def test_rotating_cache_holds_the_tokens_the_reference_holds(name):
    ref = G[f"held.{name}"]
    n_prompt = int(G[f"held.{name}.n_prompt"])
    c = O.RotatingKVCache(MAXS, keep=KEEP)
    tag = lambda a, b: torch.arange(a, b, dtype=torch.float32).reshape(1, 1, b - a, 1)     # noqa: E731
    k, _ = c.update_and_fetch(tag(0, n_prompt), tag(0, n_prompt))
    rows = [np.sort(k.reshape(-1).numpy().astype(np.int64))]
    for t in range(n_prompt, n_prompt + ref.shape[0] - 1):
        k, _ = c.update_and_fetch(tag(t, t + 1), tag(t, t + 1))
        rows.append(np.sort(k.reshape(-1).numpy().astype(np.int64)))
    for r, got in enumerate(rows):
        want = ref[r][ref[r] >= 0]
        assert np.array_equal(got, want), (name, r, got[:8], want[:8])
    assert [c.size(), c.offset] == G[f"held.{name}.size_offset"].tolist()


def test_rotating_cache_known_answers():
    c = O.RotatingKVCache(MAXS, keep=KEEP)
    z = lambda n: torch.zeros(1, 1, n, 2)      # noqa: E731
    c.update_and_fetch(z(10), z(10))
    ka = [int(c.is_trimmable()), c.trim(3), c.offset, c.size()]
    for _ in range(20):
        c.update_and_fetch(z(1), z(1))
    ka += [int(c.is_trimmable()), c.offset, c.size()]
    assert ka == G["class.known_answers"].tolist()


@pytest.fixture(scope="module")
def tiny():
    cfg = oq.tiny_cfg()
    return cfg, oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)


@pytest.mark.parametrize("name", ["short", "exact", "long"])
def test_teacher_forced_decode_over_the_rotating_cache_equals_the_reference(tiny, name):
    cfg, W = tiny
    got = oq.decode_teacher_forced(W, cfg, G[f"tf.{name}.input_ids"], None, None, G["tf.forced"], max_kv_size=MAXS, rope_mode="fallback")
    ref = G[f"tf.{name}.logits"]
    assert got.shape == ref.shape
    assert np.array_equal(got.float().numpy(), ref), np.abs(got.float().numpy() - ref).max(axis=-1)
    # ... and it is a different function from the unbounded cache from the step the reference's parts (the position quirk
    # and the dropped tokens): the same rows equal the plain decode as in the reference's own run
    plain = oq.decode_teacher_forced(W, cfg, G[f"tf.{name}.input_ids"], None, None, G["tf.forced"], rope_mode="fallback")
    same = [bool(torch.equal(a, b)) for a, b in zip(got, plain)]
    assert same == G[f"tf.{name}.equals_unbounded"].tolist()


@pytest.mark.parametrize("name", ["short", "long"])
def test_generate_step_with_max_kv_size_equals_the_reference(name):
    cfgp = oq.tiny_cfg()
    cfgp.text.tie_word_embeddings = False
    Wp = oq.random_weights(cfgp, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    for k in list(Wp):
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            Wp[k] = (Wp[k].float() * 0.5).to(BF)
    Wp = oq.peak_head(Wp, cfgp, gamma=4.0, stride=389, n_cycle=1000)
    toks, lps = oq.generate_greedy(Wp, cfgp, G[f"gen.{name}.input_ids"], max_tokens=len(G[f"gen.{name}.tokens"]), max_kv_size=MAXS,
                                   return_logprobs=True, rope_mode="fallback")
    assert toks == G[f"gen.{name}.tokens"].tolist()
    assert np.array_equal(lps.float().numpy(), G[f"gen.{name}.logprobs"])
