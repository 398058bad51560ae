"""max_kv_size on the paged pool, host side (no GPU): PagedSequence keeps the reference's RotatingKVCache window as a SET of
tokens in arbitrary slots - the engine writes the step's token at slot = entries held and attends over one more - and
`rotate_plan` names the moves (vlm_kv_move_tokens) that make room.  Replaying the plans on a slot -> token table must leave, after
every step, exactly the tokens the reference's class holds (tests/golden/rotating_ref.npz `held.*`: its cache.py run on keys
tagged with their token index), with the ring index (`_idx`, the rope offset of the reference's Qwen2-VL) tracked alongside."""
import os

import numpy as np
import pytest

from mlx_vlm_amd.models import cache as C

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotating_ref.npz"))
MAXS, KEEP = int(G["max_kv_size"]), int(G["keep"])


def _seq(max_tokens=4096):
    pool = C.KVPool(n_layers=1, n_kv_heads=1, head_dim=8, max_tokens=max_tokens, max_seqs=2, device="cpu", layout="paged")
    s = C.PagedSequence(pool)
    return s


@pytest.mark.parametrize("name", ["short", "exact", "long"])
def test_rotate_plans_leave_the_tokens_the_reference_holds(name):
    ref = G[f"held.{name}"]
    n_prompt = int(G[f"held.{name}.n_prompt"])
    s = _seq()
    s.set_rotating(MAXS, keep=KEEP)
    s.reserve(n_prompt + 2)
    slots = {i: i for i in range(n_prompt)}                 # slot -> token index, as the prefill writes them
    s.offset += n_prompt
    s.note_prefill(n_prompt)
    assert sorted(slots.values()) == ref[0][ref[0] >= 0].tolist()
    # what the oracle's restatement says `_idx` is at each call (pinned to the reference in test_oracle_ref_golden_rotating)
    from oracle import ops as O
    import torch
    oc = O.RotatingKVCache(MAXS, keep=KEEP)
    oc.update_and_fetch(torch.zeros(1, 1, n_prompt, 1), torch.zeros(1, 1, n_prompt, 1))
    for step in range(1, ref.shape[0]):
        assert s.rope_offset == oc._idx, (name, step)        # read BEFORE the window makes room, as the reference's forward does
        plan = s.rotate_plan()
        if plan:
            src, dst = plan
            assert not (set(src) & set(dst))
            moved = {d: slots[a] for a, d in zip(src, dst)}
            slots.update(moved)
        # the engine: token `offset` goes to slot = entries held; the step attends over slots [0, held]
        slots[s.kv_entries] = s.offset
        seen = sorted(slots[i] for i in range(s.kv_entries + 1))
        s.offset += 1
        s.note_decode_step()
        oc.update_and_fetch(torch.zeros(1, 1, 1, 1), torch.zeros(1, 1, 1, 1))
        want = ref[step][ref[step] >= 0].tolist()
        assert seen == want, (name, step, seen[:8], want[:8])
        assert s.kv_entries <= MAXS and len(s.pages) <= (max(n_prompt, MAXS) + 1 + 63) // 64 + 1
    assert [min(s.offset, MAXS), s.offset] == G[f"held.{name}.size_offset"].tolist()


def test_rotating_facade_known_answers_and_refusals():
    s = _seq()
    c = C.KVCache(s, 0)
    s.set_rotating(MAXS, keep=KEEP)
    s.offset += 10
    s.note_prefill(10)
    ka = [int(c.is_trimmable()), c.trim(3), c.offset, c.size()]
    for _ in range(20):
        s.rotate_plan()
        s.offset += 1
        s.note_decode_step()
    ka += [int(c.is_trimmable()), c.offset, c.size()]
    assert ka == G["class.known_answers"].tolist()
    assert c.max_size == MAXS and c.keep == KEEP
    with pytest.raises(NotImplementedError):
        c.trim(1)                                            # the window has wrapped
    with pytest.raises(NotImplementedError):
        s.note_prefill(5)                                    # a second multi-token update
    with pytest.raises(ValueError):
        _seq().set_rotating(5, keep=4)
    # make_prompt_cache(model, max_kv_size) (reference cache.py:45-70): keep = 4 on the shared sequence

    class LM:
        def make_cache(self):
            sq = _seq()
            return [C.KVCache(sq, i) for i in range(3)]

    pc = C.make_prompt_cache(LM(), max_kv_size=33)
    assert pc[0]._seq.rotating and pc[2].max_size == 33 and pc[1].keep == 4
    assert not C.make_prompt_cache(LM())[0]._seq.rotating


@pytest.mark.parametrize("seed", range(6))
def test_rotate_plans_against_the_oracle_for_many_windows(seed):
    """Property check over window sizes and prompt lengths the golden file does not hold (the oracle's RotatingKVCache is
    pinned to the reference on the golden cases): the tokens the pool holds for the step, and the ring index, at every step."""
    import torch

    from oracle import ops as O

    rng = np.random.default_rng(100 + seed)
    for _ in range(12):
        M = int(rng.integers(6, 200))
        L = int(rng.integers(1, 3 * M))
        steps = int(rng.integers(1, 2 * M + 20))
        oc = O.RotatingKVCache(M, keep=4)
        tag = lambda a, b: torch.arange(a, b, dtype=torch.float32).reshape(1, 1, b - a, 1)     # noqa: E731
        oc.update_and_fetch(tag(0, L), tag(0, L)) if L > 1 else oc.update_and_fetch(tag(0, 1), tag(0, 1))
        s = _seq(max_tokens=65536)
        s.set_rotating(M, keep=4)
        s.reserve(L + 2)
        slots = {i: i for i in range(L)}
        s.offset += L
        s.note_prefill(L)
        if L == 1:
            # (a one-token "prompt" goes through the reference's in-place path: same state - one entry, index 1)
            assert oc._idx == 1 and oc.offset == 1
        for t in range(L, L + steps):
            assert s.rope_offset == oc._idx, (M, L, t)
            plan = s.rotate_plan()
            if plan:
                assert not (set(plan[0]) & set(plan[1]))
                slots.update({d: slots[a] for a, d in zip(*plan)})
            slots[s.kv_entries] = t
            seen = sorted(slots[i] for i in range(s.kv_entries + 1))
            s.offset += 1
            s.note_decode_step()
            k, _ = oc.update_and_fetch(tag(t, t + 1), tag(t, t + 1))
            assert seen == sorted(int(x) for x in k.reshape(-1)), (M, L, t)
            assert s.kv_entries <= M
