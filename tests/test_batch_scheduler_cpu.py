"""The continuous-batching scheduler (mlx_vlm_amd/batch.py) driven by a mock engine on CPU - the way the reference
tests its own `BatchGenerator` (tests/test_generate.py:50-166 mock model / processor, 529-1170 scheduling and stats).

The mock engine is deterministic per request: the first token is a function of the prompt, every next token a function of
(previous token, context length, rope position).  Whatever the scheduler does with the rows - admit in waves, keep
prefilled requests waiting for a row, move the last row into a hole, run 1 / 2 / 4 / 8-wide steps with idle rows parked
on the scratch page - every request must emit exactly the stream it emits alone, finish by the reference's rules
(ar.py:1313-1316) and give its KV pages back."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from mlx_vlm_amd.batch import BatchGenerator
from mlx_vlm_amd.models.cache import KVPool, PagedSequence

V = 5003


def first_token(ids):
    return int((int(np.sum(ids)) * 31 + len(ids)) % V)


def next_token(tok, ctx, pos):
    return int((tok * 7 + ctx * 13 + pos * 3 + 1) % V)


def stream_alone(ids, n, delta=0):
    out, tok, ctx = [], first_token(ids), len(ids)
    for _ in range(n):
        out.append(tok)
        tok, ctx = next_token(tok, ctx, ctx + delta), ctx + 1
    return out


class MockEngineGenerator(BatchGenerator):
    """BatchGenerator with the five engine hooks replaced; everything else is the product code."""

    def __init__(self, pool, **kw):
        self.decode_widths = []
        self.prefill_sizes = []
        lm = SimpleNamespace(device="cpu", pool=pool)
        super().__init__(SimpleNamespace(language_model=lm), None, **kw)

    def _new_decode_state(self, cap):
        z = lambda: torch.zeros(cap, dtype=torch.int32)  # noqa: E731
        return SimpleNamespace(B=cap, tok=z(), pos=z(), ctx=z(), step=torch.zeros(1, dtype=torch.int32), nsplit=1,
                               last_lp=torch.zeros(cap))

    def _prefill_requests(self, batch):
        self.prefill_sizes.append(len(batch))
        caches, lens, toks = [], [], []
        for uid, ids, max_tokens, kw in batch:
            seq = PagedSequence(self.lm.pool)
            seq.reserve(len(ids) + max_tokens + 2)
            seq.offset = len(ids)
            caches.append([SimpleNamespace(_seq=seq)])
            lens.append(len(ids))
            toks.append(first_token(ids))
        tok0 = torch.tensor(toks, dtype=torch.int32)
        lp0 = -tok0.float() / V if self.compute_logprobs else None
        ctx = np.asarray(lens, dtype=np.int32)
        delta = np.asarray([b[3].get("delta", 0) for b in batch], dtype=np.int32)
        return caches, lens, tok0, lp0, torch.from_numpy(np.stack([ctx + delta, ctx]))

    def _decode_rows(self, width):
        st = self._st
        self.decode_widths.append((width, len(self._rows)))
        live = len(self._rows)
        # rows past the live ones must sit on the scratch page with a bounded context (they run inside the step)
        assert torch.all(self._table[live:] == self._scratch_seq.pages[0])
        assert int(st.ctx[live:width].max() if width > live else 0) <= 4 * 64
        for r in range(live):      # a live row's table row is its own sequence's pages
            seq = self._rows[r].seq
            assert self._table[r, : len(seq.pages)].tolist() == seq.pages
            assert int(st.ctx[r]) == seq.offset
        for r in range(width):
            t, c, p = int(st.tok[r]), int(st.ctx[r]), int(st.pos[r])
            st.tok[r] = next_token(t, c, p)
            st.last_lp[r] = -float(st.tok[r]) / V
        st.ctx[:width] += 1
        st.pos[:width] += 1

    def _row_logprobs(self, n):
        return self._st.last_lp[:n].clone()


def make_pool(paged=True, max_seqs=16):
    return KVPool(n_layers=1, n_kv_heads=1, head_dim=128, max_tokens=64 * 64, max_seqs=max_seqs, device="cpu",
                  layout="paged" if paged else "identity")


def drain(gen, on_round=None):
    got, reasons, prompts, rounds = {}, {}, {}, 0
    while gen.has_work:
        pr, out = gen.next()
        rounds += 1
        assert rounds < 2000 and len(gen) <= gen.completion_batch_size
        for p in pr:
            prompts[p.uid] = p.prompt_tokens
        seen = set()
        for r in out:
            assert r.uid not in reasons and r.uid not in seen
            seen.add(r.uid)
            got.setdefault(r.uid, []).append((r.token, r.token_logprob))
            if r.finish_reason:
                reasons[r.uid] = r.finish_reason
        if on_round:
            on_round(rounds, got)
    return got, reasons, prompts, rounds


@pytest.mark.parametrize("cap,pbs,ahead", [(4, 2, 0), (4, 2, 2), (8, 8, 2), (1, 1, 0), (3, 5, 1)])
def test_every_request_emits_its_own_stream(cap, pbs, ahead):
    rng = np.random.default_rng(cap * 10 + ahead)
    pool = make_pool()
    free_pages, free_seqs = len(pool._free_pages), len(pool._free_seqs)
    prompts = [rng.integers(1, 999, int(rng.integers(3, 40))) for _ in range(19)]
    max_tokens = [int(rng.integers(1, 24)) for _ in prompts]
    deltas = [int(rng.integers(-5, 3)) for _ in prompts]
    gen = MockEngineGenerator(pool, completion_batch_size=cap, prefill_batch_size=pbs, prefill_ahead=ahead)
    uids = gen.insert(prompts, max_tokens, prompt_kwargs=[{"delta": d} for d in deltas])
    assert [len(x[1]) for x in gen.unprocessed_prompts] == sorted(len(p) for p in prompts)      # shortest first
    got, reasons, seen_prompts, _ = drain(gen)
    for u, p, m, d in zip(uids, prompts, max_tokens, deltas):
        want = stream_alone(p, m, d)
        assert [t for t, _ in got[u]] == want, u
        np.testing.assert_allclose([lp for _, lp in got[u]], [-t / V for t in want], rtol=1e-6)
        assert reasons[u] == "length" and seen_prompts[u] == len(p)
    st = gen.stats()
    assert st.generation_tokens == sum(max_tokens) and st.prompt_tokens == sum(len(p) for p in prompts)
    assert all(w in (1, 2, 4, 8) and w >= n >= 1 and (w == 1 or w // 2 < n) for w, n in gen.decode_widths)
    assert max(gen.prefill_sizes) <= pbs
    gen.close()
    assert len(pool._free_pages) == free_pages and len(pool._free_seqs) == free_seqs       # everything given back


def test_stop_tokens_finish_with_stop_and_report_the_token():
    pool = make_pool()
    prompts = [np.arange(1, 6), np.arange(2, 12), np.arange(3, 9)]
    alone = [stream_alone(p, 10) for p in prompts]
    stop = alone[1][3]
    gen = MockEngineGenerator(pool, max_tokens=10, stop_tokens={stop}, compute_logprobs=False)
    uids = gen.insert(prompts)
    got, reasons, _, _ = drain(gen)
    for u, s in zip(uids, alone):
        want = s[: s.index(stop) + 1] if stop in s else s
        assert [t for t, _ in got[u]] == want and all(lp == 0.0 for _, lp in got[u])
        assert reasons[u] == ("stop" if want[-1] == stop else "length")
    assert reasons[uids[1]] == "stop"
    gen.close()


def test_remove_in_queue_while_prefilled_and_while_decoding():
    pool = make_pool()
    free_pages = len(pool._free_pages)
    prompts = [np.arange(1, 4 + i) for i in range(8)]
    gen = MockEngineGenerator(pool, max_tokens=12, completion_batch_size=2, prefill_batch_size=2, prefill_ahead=2)
    uids = gen.insert(prompts)
    assert gen.remove(uids[7]) and not gen.remove(uids[7])               # still queued
    early = {}
    for _ in range(2):                                                   # admits 2 (+2 ahead), rows 0, 1 run
        for r in gen.next()[1]:
            early.setdefault(r.uid, []).append((r.token, r.token_logprob))
    waiting = [u for p in gen._pending for (u, *_rest) in p.batch[p.joined:]]
    assert len(waiting) == 2 and gen.remove(waiting[0])                  # prefilled, waiting for a row
    running = [row.uid for row in gen._rows]
    assert gen.remove(running[0]) and not gen.remove(12345)              # decoding
    removed = {uids[7], waiting[0], running[0]}
    got, reasons, _, _ = drain(gen)
    got = {u: early.get(u, []) + got.get(u, []) for u in uids}
    for u, p in zip(uids, prompts):
        if u in removed:
            assert u not in reasons                      # no finish for a removed request ...
            assert [t for t, _ in got.get(u, [])] == stream_alone(p, 12)[: len(got.get(u, []))]   # ... only a prefix
            continue
        assert [t for t, _ in got[u]] == stream_alone(p, 12)
    survivors = [u for u in uids if u not in removed]
    assert sorted(reasons) == sorted(survivors)
    gen.close()
    assert len(pool._free_pages) == free_pages


def test_insert_while_running_and_identity_layout_pool():
    """Requests inserted between rounds join as rows free up; the pool layout (paged / identity) is invisible to the
    scheduler because it addresses KV through its own table."""
    pool = make_pool(paged=False, max_seqs=12)
    gen = MockEngineGenerator(pool, max_tokens=9, completion_batch_size=4, prefill_batch_size=4, prefill_ahead=1)
    first = [np.arange(1, 8), np.arange(1, 5)]
    late = [np.arange(5, 30), np.arange(2, 4), np.arange(7, 19)]
    uids = gen.insert(first)
    state = {}

    def on_round(n, got):
        if n == 3:
            state["late"] = gen.insert(late, [5, 9, 2])

    got, reasons, _, _ = drain(gen, on_round)
    for u, p, m in zip(uids + state["late"], first + late, [9, 9, 5, 9, 2]):
        assert [t for t, _ in got[u]] == stream_alone(p, m) and reasons[u] == "length"
    gen.close()


def test_rejects_what_the_built_path_does_not_do():
    pool = make_pool()
    with pytest.raises(NotImplementedError):
        MockEngineGenerator(pool, kv_bits=4)
    with pytest.raises(TypeError):
        MockEngineGenerator(pool, sampler=3)                       # neither a Sampler nor a callable
    # accepted since round 4 (the reference accepts them): a Python sampler, Python logits processors, quantized_kv_start
    # (without effect on the uniform scheme in the batch path, ar.py:776-812), thinking budgets
    MockEngineGenerator(pool, sampler=lambda lp: lp.argmax(-1)).close()
    gen = MockEngineGenerator(pool)
    with pytest.raises(ValueError):
        gen.insert([np.array([], dtype=np.int64)])
    with pytest.raises(TypeError):
        gen.insert([np.arange(3)], logits_processors=[[123]])
    with pytest.raises(ValueError):
        gen.insert([np.arange(3)], max_tokens=[1, 2])
    with pytest.raises(ValueError):
        gen.insert([np.arange(3)], thinking_budget_criteria=[None, None])
    assert len(gen.insert([np.arange(1, 4)], logits_processors=[[lambda t, l: l]])) == 1
    gen.close()


def test_thinking_budget_forces_the_next_token_of_its_row_only():
    """insert(..., thinking_budget_criteria=) (ar.py:1303-1350): every reported token of the row is shown to its criteria; a
    pending forced id replaces the row's NEXT token (already sampled by the step in flight), the other rows are untouched,
    and the forced token is what the following step is fed (the mock's chain continues from it)."""
    from mlx_vlm_amd.utils import ThinkingBudgetCriteria

    class Tok:
        def encode(self, t, add_special_tokens=False):
            return {"<think>": [V - 1], "</think>": [V - 2], "\n": [V - 3]}[t]

    pool = make_pool()
    gen = MockEngineGenerator(pool, completion_batch_size=4, prefill_batch_size=4, max_tokens=12)
    prompts = [np.arange(1, 8), np.arange(3, 14)]
    crit = ThinkingBudgetCriteria(Tok(), thinking_budget=3, thinking_start_token="<think>", enable_thinking=True,
                                  prompt_preopens_thinking=True)          # the prompt opened the block: counting from token 1
    uids = gen.insert(prompts, thinking_budget_criteria=[crit, None])
    got, reasons, _, _ = drain(gen)
    assert [t for t, _ in got[uids[1]]] == stream_alone(prompts[1], 12)       # the row without a budget: its own chain
    mine = [t for t, _ in got[uids[0]]]
    # expected: the mock chain, except that after the 4th thinking token the next two tokens are "\n", "</think>"
    exp, tok, ctx = [], first_token(prompts[0]), len(prompts[0])
    c2 = ThinkingBudgetCriteria(Tok(), thinking_budget=3, thinking_start_token="<think>", enable_thinking=True,
                                prompt_preopens_thinking=True)
    for _ in range(12):
        exp.append(tok)
        c2(tok)
        forced = c2.pop_forced_token_id()
        nxt = next_token(tok, ctx, ctx)
        tok, ctx = (forced if forced is not None else nxt), ctx + 1
    assert mine == exp and exp[4:6] == [V - 3, V - 2] and len(set(exp)) > 4
    gen.close()


def test_per_request_logits_processors_travel_with_their_rows():
    """insert(..., logits_processors=) (ar.py:2584-2606): the spec of a request reaches the row it is given, follows the
    row when the scheduler moves it into a hole, and a row re-used by a request without processors carries none; the
    streams themselves are untouched by the bookkeeping (mock engine)."""
    from mlx_vlm_amd.sample_utils import make_logits_processors

    pool = make_pool()
    gen = MockEngineGenerator(pool, completion_batch_size=2, prefill_batch_size=2, async_prefill=False)
    prompts = [np.arange(3 + i) + 5 for i in range(5)]
    specs = [make_logits_processors(repetition_penalty=1.2), None, [make_logits_processors(logit_bias={3: 1.0})], None,
             make_logits_processors(presence_penalty=0.5)]
    uids = gen.insert(prompts, [3, 7, 4, 2, 5], logits_processors=specs)
    want = {u: (s[0] if isinstance(s, list) else s) for u, s in zip(uids, specs)}
    seen = {}
    got = {u: [] for u in uids}
    while gen.has_work:
        _, out = gen.next()
        for row in gen._rows:
            seen.setdefault(row.uid, row.procs)
            assert row.procs is want[row.uid]
        for r in out:
            got[r.uid].append(r.token)
    assert set(seen) == set(uids)
    for u, p, m in zip(uids, prompts, [3, 7, 4, 2, 5]):
        assert got[u] == stream_alone(p, m)
    with pytest.raises(ValueError):
        gen.insert([np.arange(3)], logits_processors=[None, None])
    gen.close()


def test_stats_split_wall_time_into_prompt_and_decode_time():
    """prompt_time ends when a prefill is seen complete - not when its requests get a row (an admission prefilled AHEAD
    waits for rows while the others decode) - and generation_time is the wall time with decode steps in flight: the two
    add up to the wall time of the job, to within a step (reference stats: ar.py:863-884, 2705-2887)."""
    import time

    T_PRE, T_STEP, N_TOK = 0.040, 0.004, 40      # the admission prefilled ahead waits ~ N_TOK steps = 160 ms for its rows

    class Slow(MockEngineGenerator):
        def _prefill_requests(self, batch):
            time.sleep(T_PRE)
            return super()._prefill_requests(batch)

        def _decode_rows(self, width):
            time.sleep(T_STEP)
            return super()._decode_rows(width)

    pool = make_pool()
    gen = Slow(pool, completion_batch_size=2, prefill_batch_size=2, prefill_ahead=2)
    prompts = [np.arange(1, 9) + i for i in range(4)]
    gen.insert(prompts, N_TOK)
    t0 = time.perf_counter()
    got, reasons, _, rounds = drain(gen)
    wall = time.perf_counter() - t0
    st = gen.stats()
    assert len(gen.prefill_sizes) == 2 and st.generation_tokens == 4 * N_TOK
    # two prefills of T_PRE; the second one waited ~ N_TOK steps for its rows, which must not count
    assert 2 * T_PRE * 0.9 < st.prompt_time < 3 * T_PRE + 0.05, st.prompt_time      # (the bug: 2 T_PRE + 160 ms; bounds loose for a busy host)
    steps = st.decode_steps
    assert steps >= 2 * (N_TOK - 1)
    assert steps * T_STEP * 0.9 < st.generation_time < 2 * steps * T_STEP + 0.1, (st.generation_time, steps)
    assert st.prompt_time + st.generation_time < wall + 2 * T_STEP + 0.01
    assert st.prompt_time + st.generation_time > 0.8 * wall
    gen.close()


def test_wide_rows_on_request_and_pool_limits():
    """completion_batch_size > 16 asks for WIDE steps (32 / 64 rows): granted up to the language model's MAX_DECODE_ROWS and
    the pool's sequence slots (2 * rows + 2: rows + admissions prefilled ahead + the scratch sequence); every request still
    emits its own stream through 32-wide steps."""
    rng = np.random.default_rng(77)
    pool = KVPool(n_layers=1, n_kv_heads=1, head_dim=128, max_tokens=256 * 64, max_seqs=80, device="cpu", layout="paged")
    gen = MockEngineGenerator(pool, completion_batch_size=64)          # 80 < 2 * 64 + 2: halved
    assert gen.completion_batch_size == 16                              # (the mock engine has no MAX_DECODE_ROWS: 16-row default)
    gen.close()
    gen = MockEngineGenerator(pool, completion_batch_size=32, prefill_batch_size=32)
    gen.lm.MAX_DECODE_ROWS = 64
    gen.close()

    class Wide(MockEngineGenerator):
        def __init__(self, pool, **kw):
            self.decode_widths, self.prefill_sizes = [], []
            lm = SimpleNamespace(device="cpu", pool=pool, MAX_DECODE_ROWS=64)
            BatchGenerator.__init__(self, SimpleNamespace(language_model=lm), None, **kw)

    gen = Wide(pool, completion_batch_size=32, prefill_batch_size=32)
    assert gen.completion_batch_size == 32
    prompts = [rng.integers(1, 999, int(rng.integers(3, 30))) for _ in range(40)]
    max_tokens = [int(rng.integers(2, 12)) for _ in prompts]
    uids = gen.insert(prompts, max_tokens)
    got, reasons, _, _ = drain(gen)
    for u, p, m in zip(uids, prompts, max_tokens):
        assert [t for t, _ in got[u]] == stream_alone(p, m), u
    assert any(w == 32 for w, _ in gen.decode_widths)
    gen.close()
    small = KVPool(n_layers=1, n_kv_heads=1, head_dim=128, max_tokens=64 * 64, max_seqs=40, device="cpu", layout="paged")
    gen = Wide(small, completion_batch_size=64)
    assert gen.completion_batch_size == 16                              # 40 slots: 64 -> 32 -> 16 rows
    gen.close()
