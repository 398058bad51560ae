"""Uniform 8-bit KV cache on MI355X (reference QuantizedKVCache cache.py:233-334, quantized SDPA
base.py:260-302, switch-over generate/common.py:170-181 called after every forward, ar.py:362) through the engine:

  * teacher-forced decode over a cache quantised right after the prefill (quantized_kv_start = 0) and over one that
    switches in the middle of the decode (the step whose STARTING offset reaches quantized_kv_start is the first to attend
    over the 8-bit pools) - every step's logits against the oracle's typed graph of the same policy;
  * generate_step(kv_bits=8, ...) and BatchGenerator(kv_bits=8): the switch happens where the reference's would, the rows
    of a batch equal the single requests up to bf16 ties;
  * the options the built path does not cover are refused.
The operator-level checks (bit-exact quantisation, the attention kernel vs quantized_scaled_dot_product_attention) are in
tests/test_ops_gpu.py.  mx.quantize's arithmetic itself is restated from MLX's published algorithm (oracle/quant.py:
"parity unpinned" - no mlx here); the GRAPH around it (when the cache switches, what attends over what, the typed q * scale,
bf16 scores and probabilities) follows the reference's files line by line.
"""
import numpy as np
import pytest
import torch

from oracle import qwen2_vl as oq
from tests.helpers import build_product_model, synth_request

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


@pytest.fixture(scope="module")
def tiny():
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    return cfg, W, build_product_model(cfg, W, kv_pool_tokens=8192, max_seqs=24)


@pytest.mark.parametrize("start", [0, 30, 10 ** 6])
@pytest.mark.parametrize("sizes", [[(56, 84)], []])
def test_teacher_forced_decode_over_the_quantized_cache_every_step(tiny, sizes, start):
    """start = 0: quantised right after the prefill; 30 (text) / prompt + 30: the switch falls inside the 70 forced steps
    (which also cross the 64-token page boundary on both sides of it); 10**6: kv_bits given but never reached = the bf16
    path.  Every row within 2e-2 rel-rms of the oracle run with the same policy (2 layers of bf16 + 8-bit K / V)."""
    cfg, W, model = tiny
    lm = model.language_model
    ids, pix, thw = synth_request(cfg, sizes, n_text=14, seed=60 + len(sizes)) if sizes else \
        (np.random.default_rng(61).integers(3, 1000, (1, 23)), None, None)
    L0 = None
    forced = np.random.default_rng(62).integers(3, 1000, 70)
    kw = dict(image_grid_thw=thw) if thw is not None else {}
    f = model.get_input_embeddings(ids, torch.from_numpy(pix) if pix is not None else None, **kw)
    cache = lm.make_cache()
    seq = cache[0]._seq
    out = lm(ids, f.inputs_embeds, cache=cache, position_ids=f.position_ids, rope_deltas=f.rope_deltas, logits_to_keep=1)
    L0 = seq.offset
    start_abs = start if start in (0, 10 ** 6) else L0 + start
    rows = [out.logits[0, -1].clone()]

    def maybe_quantize():                      # generate_step's rule (ar.py:362, common.py:170-181)
        if not seq.q8 and seq.offset >= start_abs:
            lm.quantize_kv([seq], bits=8, group_size=64)

    maybe_quantize()
    switched_at = 0 if seq.q8 else None
    for i, y in enumerate(forced):
        rows.append(lm(np.array([[int(y)]]), cache=cache).logits[0, -1].clone())
        was = seq.q8
        maybe_quantize()
        if seq.q8 and not was:
            switched_at = i + 1
    got = torch.stack(rows)
    assert (switched_at == 0) if start == 0 else (switched_at is None) if start == 10 ** 6 else (0 < switched_at < len(forced))
    seq.release()
    ref = oq.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pix).to(BF) if pix is not None else None, thw, forced, kv_bits=8,
                                   kv_group_size=64, quantized_kv_start=start_abs)
    errs = [_rel_rms(got[i], ref[i]) for i in range(ref.shape[0])]
    assert max(errs) < 2e-2, (start, max(errs), errs)
    if start != 10 ** 6:
        # the quantised steps really ran on the 8-bit pools: they follow the quantised oracle, and that oracle differs from the
        # bf16 one by about the 8-bit noise (1 %): mean distance to the RIGHT oracle below the distance to the other one
        plain = oq.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pix).to(BF) if pix is not None else None, thw, forced)
        tail = range(max(1, (switched_at or 0) + 2), ref.shape[0])
        d_q = np.mean([_rel_rms(got[i], ref[i]) for i in tail])
        d_p = np.mean([_rel_rms(got[i], plain[i]) for i in tail])
        assert d_q < d_p, (d_q, d_p)
    print(f"quantized KV start={start} images={len(sizes)}: worst row rel-rms {max(errs):.4f}, switched at {switched_at}")


def test_generate_step_with_kv_bits_switches_where_the_reference_does(tiny):
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    ids = np.random.default_rng(71).integers(3, 1000, (1, 40))
    pool = model.language_model.pool
    # start beyond the run: identical to no kv_bits at all
    a = [(t, lp.float().cpu()) for t, lp in generate_step(ids, model, None, None, max_tokens=12, temperature=0.0)]
    b = [(t, lp.float().cpu()) for t, lp in generate_step(ids, model, None, None, max_tokens=12, temperature=0.0, kv_bits=8,
                                                           quantized_kv_start=5000)]
    assert [t for t, _ in a] == [t for t, _ in b] and all(torch.equal(x[1], y[1]) for x, y in zip(a, b))
    # start inside the run: the first steps are bit-identical to the plain run, the later ones differ (8-bit K / V)
    c = [(t, lp.float().cpu()) for t, lp in generate_step(ids, model, None, None, max_tokens=12, temperature=0.0, kv_bits=8,
                                                           quantized_kv_start=44, lookahead=3)]
    # forwards: prefill (offset 40), steps 1..: the cache is at 44 after the 4th decode forward -> generated tokens 0..4 (the
    # first token + four steps) come from the bf16 cache
    for i in range(5):
        assert c[i][0] == a[i][0] and torch.equal(c[i][1], a[i][1]), i
    assert pool.kpool8 is not None
    assert any(not torch.equal(c[i][1], a[i][1]) for i in range(5, 12))
    with pytest.raises(NotImplementedError):
        next(generate_step(ids, model, None, None, max_tokens=2, kv_bits=4))


def test_generate_step_with_kv_bits_and_a_chunked_prompt(tiny):
    """prefill_step_size (default 2048) feeds a long prompt in chunks; with kv_bits the cache must not turn 8-bit
    between the chunks (the chunk path attends over bf16 pages).  kv_bits = 8, quantized_kv_start = 0 and a prompt of several
    small chunks: the run completes, the cache IS 8-bit afterwards, and the tokens / log-probs equal the one-chunk prompt's
    (the prompt pass sees unrounded K / V either way; every decode step attends over the same 8-bit pools)."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    ids = np.random.default_rng(72).integers(3, 1000, (1, 90))
    kw = dict(max_tokens=8, temperature=0.0, kv_bits=8, quantized_kv_start=0)
    whole = [(t, lp.float().cpu()) for t, lp in generate_step(ids, model, None, None, prefill_step_size=4096, **kw)]
    chunked = [(t, lp.float().cpu()) for t, lp in generate_step(ids, model, None, None, prefill_step_size=32, **kw)]
    assert model.language_model.pool.kpool8 is not None and len(chunked) == len(whole) == 8
    # tokens equal up to the first bf16 tie (chunked attention sums in another order), the chosen tokens' log-probs close
    n_equal = 0
    for (ta, la), (tb, lb) in zip(whole, chunked):
        assert abs(float(la[ta]) - float(lb[ta])) <= 0.13
        if ta != tb:
            break
        n_equal += 1
    assert n_equal >= 4, (n_equal, [t for t, _ in whole], [t for t, _ in chunked])


def test_batch_generator_with_kv_bits_equals_single_requests(tiny):
    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.generate import generate_step
    from tests.test_engine_gpu import _assert_streams_equal_up_to_ties, _mixed_requests

    cfg, W, model = tiny
    reqs = _mixed_requests(cfg, 7, seed0=340)
    max_tokens = [6 + (5 * i) % 9 for i in range(7)]
    singles = []
    for (ids, pix, thw), m in zip(reqs, max_tokens):
        kw = dict(image_grid_thw=thw) if thw is not None else {}
        singles.append([(t, float(lp[t])) for t, lp in generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None,
                                                                     None, max_tokens=m, kv_bits=8, quantized_kv_start=0, **kw)])
    gen = BatchGenerator(model, None, max_tokens=8, completion_batch_size=4, prefill_batch_size=2, kv_bits=8, quantized_kv_start=0)
    pk = [dict(pixel_values=torch.from_numpy(p), image_grid_thw=g) if p is not None else {} for _, p, g in reqs]
    uids = gen.insert([r[0].reshape(-1) for r in reqs], list(max_tokens), prompt_kwargs=pk)
    got = {u: [] for u in uids}
    while gen.has_work:
        _, out = gen.next()
        assert all(row.seq.q8 for row in gen._rows)
        for r in out:
            got[r.uid].append((r.token, r.token_logprob))
    gen.close()
    _assert_streams_equal_up_to_ties([got[u] for u in uids], singles)
    # (quantized_kv_start > 0 is accepted since round 4: per-row switch-over, tests/test_batch_api_gpu.py)


def test_batch_policy_keeps_the_last_layer_of_a_deep_stack_unquantized():
    """The reference's BATCH policy (models/cache.py:8-21 should_quantize_kv_layer, generate/ar.py:842-858; known answers in
    its tests/test_batch_quantized_cache.py:256-276 and in tests/golden/kvquant_ref.npz): with kv_bits the last layer of a
    stack deeper than 2 keeps its unquantised cache.  A 3-layer model, 4 rows through the batch step's forward
    (decode_forward_rows sets vlm_kv_pool.q8_skip_last): every row of 5 steps follows the oracle with layers 0-1 quantised
    and layer 2 in bf16, and is FARTHER from the all-layers-quantised graph than from that one."""
    cfg = oq.tiny_cfg()
    cfg.text.num_hidden_layers = 3
    W = oq.random_weights(cfg, seed=4321, dtype=BF, std=0.05, embed_std=0.2)
    model = build_product_model(cfg, W, kv_pool_tokens=8192, max_seqs=24)
    lm = model.language_model
    B, steps = 4, 5
    rng = np.random.default_rng(88)
    prompts = [rng.integers(3, 1000, n).astype(np.int64) for n in (70, 9, 130, 33)]
    forced = rng.integers(3, 1000, (steps, B))
    caches = lm.make_cache_batch(B)
    for p, c in zip(prompts, caches):
        lm(p[None], cache=c, logits_to_keep=1)
    seqs = [c[0]._seq for c in caches]
    lm.quantize_kv(seqs, bits=8, group_size=64)
    st = lm.decode_begin(caches, forced[0], np.zeros((B, 1), dtype=np.int64), max_new_tokens=steps + 1)
    table = lm.pool.block_table[st.seq_row0:]
    rows = []
    for s in range(steps):
        st.tok[:B].copy_(torch.from_numpy(forced[s].astype(np.int32)).cuda())
        lm.decode_forward_rows(st, B, table, q8=True)
        rows.append(st.logits[:B].clone())
        lm.decode_advance_rows(st, B)
        for sq in seqs:
            sq.offset += 1
    got = torch.stack(rows)
    for sq in seqs:
        sq.release()
    d_batch, d_all = [], []
    for b in range(B):
        ref = oq.decode_teacher_forced(W, cfg, prompts[b][None], None, None, forced[:, b], kv_bits=8, quantized_kv_start=0,
                                       kv_batch_policy=True)[1:]
        ref_all = oq.decode_teacher_forced(W, cfg, prompts[b][None], None, None, forced[:, b], kv_bits=8, quantized_kv_start=0)[1:]
        for s in range(steps):
            e = _rel_rms(got[s, b], ref[s])
            assert e < 2e-2, (b, s, e)
            d_batch.append(e)
            d_all.append(_rel_rms(got[s, b], ref_all[s]))
    assert np.mean(d_batch) < np.mean(d_all), (np.mean(d_batch), np.mean(d_all))
    print(f"batch kv policy: mean distance to the last-layer-bf16 graph {np.mean(d_batch):.4f}, to the all-quantised one {np.mean(d_all):.4f}")
