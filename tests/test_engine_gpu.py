"""GPU parity tests, model level: the native ViT / prefill / decode engines and
the Python API (generate_step, stream_generate, batch) against the oracle
(oracle/qwen2_vl.py) on the same seeded bf16 weights and inputs, plus the HF
goldens through the HIP path.

Tolerances: activations are bf16 with ~1-ulp differences per op that compound
over depth; logits are compared with an absolute tolerance relative to the logit
rms (stated per test); greedy tokens must be identical unless the oracle's own
top-2 logits are closer than that tolerance (tie-aware check, written out below).
"""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import qwen2_vl as oq
from tests.helpers import bf16_close, build_product_model, synth_request

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen2_vl_tiny_hf.npz"))


@pytest.fixture(scope="module")
def tiny():
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    model = build_product_model(cfg, W, kv_pool_tokens=8192, max_seqs=16)
    return cfg, W, model


def _rel_rms_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def test_vision_tower_vs_oracle(tiny):
    cfg, W, model = tiny
    _, pix, thw = synth_request(cfg, [(56, 84), (112, 56)], seed=3)
    ref = oq.vision_tower(W, cfg, torch.from_numpy(pix).to(BF), thw)
    out = model.vision_tower(torch.from_numpy(pix), thw)
    assert out.shape == ref.shape
    e = _rel_rms_err(out, ref)
    assert e < 2e-2, e      # bf16 through 2 blocks + merger: ~1% rms


def test_vision_tower_vs_hf_golden_fp32_weights(tiny):
    """HF fp32 features vs the HIP bf16 path: bf16 weights/activations => ~1-2% rms."""
    cfg, _, _ = tiny
    W32 = oq.random_weights(cfg, seed=1234, dtype=torch.float32, std=0.05, embed_std=0.2)
    model = build_product_model(cfg, W32, kv_pool_tokens=2048, max_seqs=4)
    for case in ("one_image", "two_images"):
        out = model.vision_tower(torch.from_numpy(G[case + ".pixel_values"]), G[case + ".grid_thw"])
        e = _rel_rms_err(out, torch.from_numpy(G[case + ".hf_image_features"]))
        assert e < 3e-2, (case, e)


def test_rope_index_product_vs_golden(tiny):
    cfg, _, model = tiny
    for case in ("one_image", "two_images"):
        pos, deltas = model.language_model.get_rope_index(G[case + ".input_ids"], G[case + ".grid_thw"])
        np.testing.assert_array_equal(pos, G[case + ".hf_position_ids"])
        np.testing.assert_array_equal(deltas, G[case + ".hf_rope_deltas"])


def _oracle_prefill_logits(W, cfg, ids, pix, thw):
    emb, pos, deltas = oq.get_input_embeddings(W, cfg, ids, torch.from_numpy(pix).to(BF) if pix is not None else None, thw)
    cache = [O.KVCache() for _ in range(cfg.text.num_hidden_layers)]
    h = oq.qwen2_model(W, cfg, emb, cache, torch.from_numpy(np.asarray(pos)))
    return oq.lm_head(W, cfg, h)[0], cache, deltas


def test_prefill_logits_and_kv_vs_oracle(tiny):
    cfg, W, model = tiny
    ids, pix, thw = synth_request(cfg, [(56, 84)], n_text=20, seed=5)
    ref_logits, ref_cache, _ = _oracle_prefill_logits(W, cfg, ids, pix, thw)
    lm = model.language_model
    f = model.get_input_embeddings(ids, torch.from_numpy(pix), image_grid_thw=thw)
    cache = lm.make_cache()
    out = lm(ids, f.inputs_embeds, cache=cache, position_ids=f.position_ids)      # all rows (reference contract)
    assert out.logits.shape == (1, ids.shape[1], cfg.text.vocab_size)
    e = _rel_rms_err(out.logits[0], ref_logits)
    assert e < 3e-2, e
    # paged cache contents == the oracle's contiguous KVCache (facade .state)
    for layer in (0, cfg.text.num_hidden_layers - 1):
        k, v = cache[layer].state
        rk, rv = ref_cache[layer].state
        assert k.shape == rk.shape
        assert _rel_rms_err(k, rk) < 2e-2 and _rel_rms_err(v, rv) < 2e-2
    assert cache[0].offset == ids.shape[1]
    cache[0]._seq.release()


def test_logits_to_keep_last_row_equals_all_rows_last(tiny):
    cfg, W, model = tiny
    ids, pix, thw = synth_request(cfg, [(56, 56)], n_text=9, seed=6)
    lm = model.language_model
    f = model.get_input_embeddings(ids, torch.from_numpy(pix), image_grid_thw=thw)
    c1, c2 = lm.make_cache(), lm.make_cache()
    a = lm(ids, f.inputs_embeds.clone(), cache=c1, position_ids=f.position_ids).logits[:, -1]
    b = lm(ids, f.inputs_embeds.clone(), cache=c2, position_ids=f.position_ids, logits_to_keep=1).logits[:, -1]
    assert torch.equal(a, b)
    c1[0]._seq.release(); c2[0]._seq.release()


def _tie_aware_equal(toks, ref_toks, ref_logits, tol):
    """tokens equal, or the first divergence happens where the oracle's logit margin between the two
    candidates is below tol * rms (after a divergence the sequences legitimately differ)."""
    for n, (a, b) in enumerate(zip(toks, ref_toks)):
        if a == b:
            continue
        row = ref_logits[n].float()
        margin = abs(float(row[a]) - float(row[b]))
        return margin <= tol * float(row.pow(2).mean().sqrt()), n, margin
    return True, None, None


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("sizes", [[(56, 84)], [(56, 56), (84, 56)], []])
def test_greedy_generate_vs_oracle(tiny, sizes, use_graph):
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    ids, pix, thw = synth_request(cfg, sizes, n_text=14, seed=7 + len(sizes)) if sizes else \
        (np.random.default_rng(9).integers(3, 1000, (1, 17)), None, None)
    n_new = 24
    ref_toks, ref_logits = oq.generate_greedy(W, cfg, ids, torch.from_numpy(pix).to(BF) if pix is not None else None,
                                              thw, max_tokens=n_new, return_logits=True)
    kw = dict(image_grid_thw=thw) if thw is not None else {}
    gen = generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None, None, max_tokens=n_new,
                        temperature=0.0, use_graph=use_graph, lookahead=3, **kw)
    toks, lps = [], []
    for t, lp in gen:
        toks.append(t)
        lps.append(lp.float().cpu())
    assert len(toks) == n_new
    ok, n, margin = _tie_aware_equal(toks, ref_toks, ref_logits, tol=3e-2)
    assert ok, (toks, ref_toks, n, margin)
    # logprobs of the first token vs oracle: 2 bf16 ulps (|logprob| ~ 10-16 => ulp 0.06-0.125) + 3% of the rms
    ref_lp0 = O.logprobs_from_logits(ref_logits[0][None])[0]
    ok, rep = bf16_close(lps[0], ref_lp0, ulps=2, atol_rms=3e-2)
    assert ok, rep


def test_hf_golden_greedy_through_hip_path(tiny):
    """End to end against HuggingFace (fp32) tokens: bf16 HIP path, tie-aware at 5% of the logit rms."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    for case in ("one_image", "two_images"):
        ids = G[case + ".input_ids"]
        toks = [t for t, _ in generate_step(ids, model, torch.from_numpy(G[case + ".pixel_values"]), None, max_tokens=8,
                                            image_grid_thw=G[case + ".grid_thw"], return_logprobs=False)]
        ref = G[case + ".hf_greedy"].tolist()
        hf_last = torch.from_numpy(G[case + ".hf_logits"][-1])
        if toks != ref:
            n = next(i for i, (a, b) in enumerate(zip(toks, ref)) if a != b)
            assert n > 0 or abs(float(hf_last[toks[0]] - hf_last[ref[0]])) <= 5e-2 * float(hf_last.pow(2).mean().sqrt()), (toks, ref)


def test_module_contract_decode_call_matches_fused_graph(tiny):
    """language_model(y, cache=...) at L == 1 (reference contract) produces the same logits as the fused step."""
    cfg, W, model = tiny
    lm = model.language_model
    ids = np.random.default_rng(11).integers(3, 1000, (1, 21))
    cache = lm.make_cache()
    out = lm(ids, cache=cache, logits_to_keep=1)
    tok = int(O.argmax_first(O.logprobs_from_logits(out.logits[:, -1].cpu()))[0])
    toks = [tok]
    for _ in range(5):
        out = lm(np.array([[toks[-1]]]), cache=cache)
        toks.append(int(O.argmax_first(O.logprobs_from_logits(out.logits[:, -1].cpu()))[0]))
    cache[0]._seq.release()
    from mlx_vlm_amd.generate import generate_step

    fused = [t for t, _ in generate_step(ids, model, None, None, max_tokens=6, return_logprobs=False)]
    assert toks == fused


def test_stream_generate_bypass_and_stats(tiny):
    from mlx_vlm_amd.generate import stream_generate

    cfg, W, model = tiny
    ids, pix, thw = synth_request(cfg, [(56, 56)], n_text=10, seed=12)
    res = list(stream_generate(model, None, input_ids=ids, pixel_values=torch.from_numpy(pix), mask=None,
                               image_grid_thw=thw, max_tokens=12))
    last = res[-1]
    assert last.finish_reason == "length" and last.generation_tokens == 12
    assert last.prompt_tokens == ids.size and last.total_tokens == ids.size + 12
    assert last.prompt_tps > 0 and last.generation_tps > 0 and last.peak_memory > 0
    assert len(res) == 13   # one per token + the final summary (reference dispatch.py:1028-1075)


@pytest.mark.parametrize("B", [2, 3, 8])
def test_batch_generate_ids_matches_single(tiny, B):
    """Batched varlen prefill + batched decode rows == the same requests run one at a time."""
    from mlx_vlm_amd.generate import batch_generate_ids, generate_step

    cfg, W, model = tiny
    reqs = [synth_request(cfg, [(56, 56 + 28 * (i % 2))] if i % 3 != 2 else [], n_text=8 + i, seed=20 + i) if i % 3 != 2
            else (np.random.default_rng(30 + i).integers(3, 1000, (1, 11 + i)), None, None) for i in range(B)]
    singles = []
    for ids, pix, thw in reqs:
        kw = dict(image_grid_thw=thw) if thw is not None else {}
        singles.append([t for t, _ in generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None, None,
                                                    max_tokens=10, return_logprobs=False, **kw)])
    toks, stats = batch_generate_ids(model, [r[0].reshape(-1) for r in reqs],
                                     [torch.from_numpy(r[1]) if r[1] is not None else None for r in reqs],
                                     [r[2] for r in reqs], max_tokens=10)
    assert stats.generation_tokens == 10 * B
    mism = sum(int(a != b) for a, b in zip(toks, singles))
    # batched GEMV (M=2/4/8) and varlen GEMM tiles accumulate in the same order per row -> identical tokens
    assert mism == 0, (toks, singles)


@pytest.mark.parametrize("sizes", [[(56, 84)], []])
def test_decode_step_tuning_variants_are_bit_identical(tiny, sizes):
    """The fused greedy tail changes the launch structure only: tokens AND every step's log-probs are bit-identical to the
    step with the separate sampler launches."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    lm = model.language_model
    ids, pix, thw = synth_request(cfg, sizes, n_text=14, seed=31) if sizes else \
        (np.random.default_rng(32).integers(3, 1000, (1, 19)), None, None)
    kw = dict(image_grid_thw=thw) if thw is not None else {}

    def run():
        toks, lps = [], []
        for t, lp in generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None, None, max_tokens=70,
                                   temperature=0.0, lookahead=5, **kw):
            toks.append(t)
            lps.append(lp.clone())
        return toks, torch.stack(lps)

    try:
        lm.apply_tuning(fused_tail=0)
        base_t, base_lp = run()
        lm.apply_tuning(fused_tail=1)
        t, lp = run()
        assert t == base_t
        assert torch.equal(lp, base_lp)
    finally:
        lm.apply_tuning()


@pytest.mark.parametrize("use_graph", [True, False])
def test_generate_step_with_penalties_matches_oracle(tiny, use_graph):
    """repetition / presence / frequency penalties + logit_bias through generate_step (device-side logits pass inside the
    captured step, history = prompt + fed tokens) against the oracle's restatement of ar.py:360-364 - the tiny model's
    greedy fixed point is broken up by the penalties, so the token stream is not degenerate."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    ids = np.random.default_rng(61).integers(3, 1000, (1, 19))
    kw = dict(repetition_penalty=1.4, repetition_context_size=16, presence_penalty=0.6, frequency_penalty=0.3,
              logit_bias={7: 1.5, 11: -2.0})
    n_new = 30
    ref_toks, ref_logits = oq.generate_greedy(W, cfg, ids, max_tokens=n_new, return_logits=True,
                                              processors=dict(logit_bias=kw["logit_bias"], repetition_penalty=1.4,
                                                              repetition_context_size=16, presence_penalty=0.6,
                                                              frequency_penalty=0.3))
    toks = [t for t, _ in generate_step(ids, model, None, None, max_tokens=n_new, temperature=0.0, use_graph=use_graph, **kw)]
    assert len(set(ref_toks)) > 5                      # the penalties do break the fixed point
    ok, n, margin = _tie_aware_equal(toks, ref_toks, ref_logits, tol=3e-2)
    assert ok, (toks, ref_toks, n, margin)
    plain = [t for t, _ in generate_step(ids, model, None, None, max_tokens=n_new, temperature=0.0)]
    assert plain != toks


def test_prefill_onto_non_empty_cache_chunked_and_multi_turn(tiny):
    """A prompt fed in two chunks (reference ar.py:426-472) and a follow-up turn appended to a `prompt_cache` that already
    holds prompt + generated tokens (dispatch.py:861-882): the second prefill attends to [cached | new].  Against the
    oracle's one-shot prefill of the concatenation (last-row logits, 3e-2 rel-rms; cache contents 2e-2) and then decoding
    on: greedy tokens after the chunked prefill equal those after the one-shot prefill."""
    cfg, W, model = tiny
    lm = model.language_model
    ids, pix, thw = synth_request(cfg, [(56, 84)], n_text=30, seed=70)
    f = model.get_input_embeddings(ids, torch.from_numpy(pix), image_grid_thw=thw)
    L = ids.shape[1]
    cut = L - 17                                                    # the image block lies in the first chunk
    ref_logits, ref_cache, _ = _oracle_prefill_logits(W, cfg, ids, pix, thw)
    pos = np.asarray(f.position_ids)
    cache = lm.make_cache()
    emb = f.inputs_embeds.reshape(L, -1)
    # (prefill uses its input as the residual stream: every call gets its own copy)
    lm.prefill(emb[:cut].clone(), pos.reshape(3, L)[:, :cut], [cache], [cut], "last")
    assert cache[0].offset == cut
    logits = lm.prefill(emb[cut:].clone(), pos.reshape(3, L)[:, cut:], [cache], [L - cut], "last")
    assert cache[0].offset == L
    assert _rel_rms_err(logits[0], ref_logits[-1]) < 3e-2
    for layer in (0, cfg.text.num_hidden_layers - 1):
        k, v = cache[layer].state
        rk, rv = ref_cache[layer].state
        assert _rel_rms_err(k, rk) < 2e-2 and _rel_rms_err(v, rv) < 2e-2
    # one-shot cache of the same prompt: the next decode step's logits agree (the caches are interchangeable)
    one = lm.make_cache()
    a = lm.prefill(emb.clone(), pos.reshape(3, L), [one], [L], "last")
    assert _rel_rms_err(logits[0], a[0]) < 1e-2
    lm._rope_deltas = np.asarray(f.rope_deltas)
    x = lm(np.array([[77]]), cache=cache).logits[0, -1]
    y = lm(np.array([[77]]), cache=one).logits[0, -1]
    assert _rel_rms_err(x, y) < 1e-2
    cache[0]._seq.release(); one[0]._seq.release()


@pytest.mark.parametrize("prefix,n", [(300, 1), (300, 64), (2100, 37), (130, 65), (700, 200)])
def test_prompt_chunk_onto_a_cache_three_attention_forms_agree(tiny, monkeypatch, prefix, n):
    """LanguageModel._prefill_onto_cache (chunked prefill / a conversation turn) has three attention forms since round 6: up to
    64 new tokens run the paged DECODE attention (every new token a decode row over the pages, no gather of the prefix); longer
    chunks gather the prefix and run the prompt kernel with the query blocks starting at the cache length (q_start); the form of
    rounds 3-5 gave the prefix rows zero queries (VLM_ONTO_CACHE_QSTART=0).  All three against the one-shot prefill of the whole
    sequence: the chunk's last-row logits and the NEXT decode step's logits (which reads the K / V the chunk wrote)."""
    cfg, W, model = tiny
    lm = model.language_model
    L = prefix + n
    rng = np.random.default_rng(prefix * 131 + n)
    emb = lm._w["embed"][torch.from_numpy(rng.integers(3, 1000, L)).cuda()]
    pos = np.broadcast_to(np.arange(L, dtype=np.int64)[None], (3, L)).copy()
    lm._rope_deltas = np.zeros((1, 1), dtype=np.int64)
    one = lm.make_cache()
    want = lm.prefill(emb.clone(), pos, [one], [L], "last")
    want_next = lm(np.array([[77]]), cache=one).logits[0, -1].clone()
    one[0]._seq.release()
    for form, env in (("decode", {}), ("q_start", {"VLM_ONTO_CACHE_DECODE_ATTN": "0"}),
                      ("zero_queries", {"VLM_ONTO_CACHE_DECODE_ATTN": "0", "VLM_ONTO_CACHE_QSTART": "0"})):
        for k in ("VLM_ONTO_CACHE_DECODE_ATTN", "VLM_ONTO_CACHE_QSTART"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = lm.make_cache()
        lm.prefill(emb[:prefix].clone(), pos[:, :prefix], [c], [prefix], "last", reserve_extra=n + 4)
        got = lm.prefill(emb[prefix:].clone(), pos[:, prefix:], [c], [n], "last", reserve_extra=4)
        assert c[0].offset == L
        assert _rel_rms_err(got[0], want[0]) < 1e-2, (form, _rel_rms_err(got[0], want[0]))
        nxt = lm(np.array([[77]]), cache=c).logits[0, -1]
        assert _rel_rms_err(nxt, want_next) < 1e-2, (form, "next step", _rel_rms_err(nxt, want_next))
        c[0]._seq.release()


def test_generate_step_prefill_step_size_chunks_and_the_models_veto(tiny, monkeypatch):
    """generate_step(prefill_step_size=16) feeds a longer prompt in chunks of 16 until one token is left (reference
    ar.py:409-470) - same greedy tokens as the one-shot prefill, log-probs of the first token within the chunk-vs-one-shot
    distance - unless the model forbids it (`no_chunked_prefill`, or a `chunked_prefill_policy` that says no:
    generate/common.py:39-74), in which case the prompt goes in whole."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    lm = model.language_model
    ids, pix, thw = synth_request(cfg, [(56, 84)], n_text=30, seed=71)
    L = ids.shape[1]
    calls = []
    real = lm.prefill

    def spy(emb, pos, caches, lengths, *a, **k):
        calls.append(list(lengths))
        return real(emb, pos, caches, lengths, *a, **k)

    monkeypatch.setattr(lm, "prefill", spy)
    run = lambda **kw: list(generate_step(ids, model, torch.from_numpy(pix), None, max_tokens=8, image_grid_thw=thw, **kw))     # noqa: E731
    one = run(prefill_step_size=None)
    assert calls == [[L]]
    calls.clear()
    chunked = run(prefill_step_size=16)
    want = [[16]] * ((L - 1) // 16) + ([[(L - 1) % 16]] if (L - 1) % 16 else []) + [[1]]
    assert calls == want, (calls, want)
    as_pairs = lambda st: [[(t, float(lp[t])) for t, lp in st]]       # noqa: E731
    _assert_streams_equal_up_to_ties(as_pairs(chunked), as_pairs(one), min_equal=0.5)
    for veto in ("attr", "policy"):
        calls.clear()
        if veto == "attr":
            monkeypatch.setattr(lm, "no_chunked_prefill", True, raising=False)
        else:
            monkeypatch.delattr(lm, "no_chunked_prefill")
            monkeypatch.setattr(model, "chunked_prefill_policy", lambda **kw: False, raising=False)
        assert [t for t, _ in run(prefill_step_size=16)] == [t for t, _ in one]
        assert calls == [[L]], (veto, calls)


def test_generate_step_default_step_size_prefills_a_long_prompt_in_one_shot(tiny, monkeypatch):
    """with the DEFAULT prefill_step_size the engine decides - a 2500-token prompt (beyond the reference's 2048)
    goes through ONE prefill call (the one-shot kernels: 2.3 x faster than chunks at 8k tokens, scripts/r06/long_prompt.py),
    an explicit step size is honoured, and beyond ONE_SHOT_PREFILL_TOKENS the default chunks again."""
    from mlx_vlm_amd import generate as G

    cfg, W, model = tiny
    lm = model.language_model
    ids = np.random.default_rng(5).integers(3, 1000, (1, 2500))
    calls = []
    real = lm.prefill

    def spy(emb, pos, caches, lengths, *a, **k):
        calls.append(list(lengths))
        return real(emb, pos, caches, lengths, *a, **k)

    monkeypatch.setattr(lm, "prefill", spy)
    toks = [t for t, _ in G.generate_step(ids, model, None, None, max_tokens=3)]
    assert calls == [[2500]] and len(toks) == 3
    calls.clear()
    list(G.generate_step(ids, model, None, None, max_tokens=3, prefill_step_size=1024))
    assert calls == [[1024], [1024], [451], [1]], calls
    calls.clear()
    monkeypatch.setattr(G, "ONE_SHOT_PREFILL_TOKENS", 2000)
    list(G.generate_step(ids, model, None, None, max_tokens=3))
    assert calls == [[2048], [451], [1]], calls


def test_stream_generate_multi_turn_prompt_cache_state_and_vision_cache(tiny):
    """Two conversation turns through stream_generate with a PromptCacheState and a VisionFeatureCache (reference
    dispatch.py:800-809,861-882): turn 2's prompt = turn 1's prompt + its answer + new text; the cached KV prefix is reused
    (cached_tokens > 0, only the suffix is prefilled onto the cache) and the image is not encoded again; the tokens equal
    those of a cold run of the same turn-2 prompt."""
    from mlx_vlm_amd.generate import PromptCacheState, stream_generate
    from mlx_vlm_amd.vision_cache import VisionFeatureCache

    cfg, W, model = tiny
    ids1, pix, thw = synth_request(cfg, [(56, 56)], n_text=9, seed=90)
    img_key = "turn-image"
    state, vcache = PromptCacheState(), VisionFeatureCache(max_size=2)
    kw = dict(pixel_values=torch.from_numpy(pix), mask=None, image_grid_thw=thw, max_tokens=6)
    calls = {"n": 0}
    tower = model.vision_tower.__class__.__call__

    def counting(self, *a, **k):
        calls["n"] += 1
        return tower(self, *a, **k)

    model.vision_tower.__class__.__call__ = counting
    try:
        r1 = list(stream_generate(model, None, image=img_key, input_ids=ids1, prompt_cache_state=state, vision_cache=vcache, **kw))
        assert calls["n"] == 1 and len(vcache) == 1 and r1[-1].cached_tokens == 0
        t1 = [r.token for r in r1[:-1]]
        assert state.cache is not None and len(state.token_ids) == ids1.shape[1] + len(t1) - 1 == state.cache[0].offset
        ids2 = np.concatenate([ids1[0], t1, np.random.default_rng(91).integers(3, 1000, 7)])[None]
        r2 = list(stream_generate(model, None, image=img_key, input_ids=ids2, prompt_cache_state=state, vision_cache=vcache, **kw))
        assert calls["n"] == 1                                   # the image was NOT encoded again
        assert r2[-1].cached_tokens == ids1.shape[1] + len(t1) - 1
        cold = list(stream_generate(model, None, input_ids=ids2, **kw))
        assert calls["n"] == 2
        assert [r.token for r in r2[:-1]] == [r.token for r in cold[:-1]]
        assert r2[-1].prompt_tokens == cold[-1].prompt_tokens == ids2.shape[1]
    finally:
        model.vision_tower.__class__.__call__ = tower
        if state.cache is not None:
            state.cache[0]._seq.release()


def test_fused_greedy_decode_hook_matches_module_call(tiny):
    """`language_model.fused_greedy_decode(inputs, cache=)` (the reference's plug point, ar.py:1015-1042): same tokens as
    the module call + host argmax, B = 1 and B = 2; returns None (= fall back) without a cache or with processors."""
    cfg, W, model = tiny
    lm = model.language_model
    assert lm.fused_greedy_decode(np.array([[5]]), cache=None) is None
    assert lm.supports_fused_greedy_logits_processors([lambda t, l: l]) is False
    ids = np.random.default_rng(81).integers(3, 1000, (1, 21))
    c1, c2 = lm.make_cache(), lm.make_cache()
    o1 = lm(ids, cache=c1, logits_to_keep=1)
    lm(ids, cache=c2, logits_to_keep=1)
    tok = int(O.argmax_first(O.logprobs_from_logits(o1.logits[:, -1].cpu()))[0])
    a, b = [tok], [tok]
    for _ in range(6):
        out = lm(np.array([[a[-1]]]), cache=c1)
        a.append(int(O.argmax_first(O.logprobs_from_logits(out.logits[:, -1].cpu()))[0]))
        t = lm.fused_greedy_decode(np.array([[b[-1]]]), cache=c2)
        assert t is not None and t.shape == (1,)
        b.append(int(t[0]))
    assert a == b
    assert lm.fused_greedy_decode(np.array([[1]]), cache=c2, logits_processors=[lambda t, l: l]) is None
    c1[0]._seq.release(); c2[0]._seq.release()


@pytest.mark.parametrize("sampler", [dict(temperature=0.8), dict(temperature=1.0, top_p=0.9), dict(temperature=0.7, top_p=0.9, min_p=0.02, top_k=50)])
def test_sampled_fused_tail_equals_gather_sample_advance(tiny, sampler):
    """The sampled step's fused tail (vlm_sample_advance: the sampler's last launch also advances ctx / pos / step, writes the
    token ring and gathers the next step's embedding row) against the three-launch form it replaces (embedding gather, vlm_sample,
    vlm_decode_advance): the same tokens and log-prob rows over 70 steps - across the 64-token page boundary and with a lookahead
    of 5 steps enqueued ahead."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    lm = model.language_model
    ids = np.random.default_rng(33).integers(3, 1000, (1, 21))

    def run():
        toks, lps = [], []
        for t, lp in generate_step(ids, model, None, None, max_tokens=70, seed=9, lookahead=5, **sampler):
            toks.append(t)
            lps.append(lp.clone())
        return toks, torch.stack(lps)

    try:
        lm.apply_tuning(fused_tail=0)
        base_t, base_lp = run()
        lm.apply_tuning(fused_tail=1)
        t, lp = run()
        assert t == base_t
        assert torch.equal(lp, base_lp)
        assert len(set(t)) >= 2                     # (a sampled stream, not one repeated token)
    finally:
        lm.apply_tuning()


def test_sampling_temperature_reproducible_and_varied(tiny):
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    ids = np.random.default_rng(13).integers(3, 1000, (1, 15))
    a = [t for t, _ in generate_step(ids, model, None, None, max_tokens=12, temperature=1.0, top_p=0.95, seed=5)]
    b = [t for t, _ in generate_step(ids, model, None, None, max_tokens=12, temperature=1.0, top_p=0.95, seed=5)]
    c = [t for t, _ in generate_step(ids, model, None, None, max_tokens=12, temperature=1.0, top_p=0.95, seed=6)]
    assert a == b and a != c


# ---- against vectors produced by the REFERENCE'S OWN files (tests/golden/make_golden_ref.py; only the .npz is read)
RG = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen2_vl_tiny_ref.npz"))


@pytest.mark.parametrize("case", ["one_image", "two_images"])
def test_reference_golden_bf16_prefill_and_greedy_through_hip_path(tiny, case):
    """bf16 reference vectors (reference code over oracle/mlx_shim): HIP image features, last-row prefill logprobs and
    8 greedy tokens.  Tolerance: 2 bf16 ulps + 3 % of the rms (fp32 accumulation order; the reference vectors use
    the pure-MLX rope path, the kernels the fused kernel's fp32 numerics - within the reference's own 1e-4 contract)."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    p = case + ".bf16."
    pix, thw, ids = G[case + ".pixel_values"], RG[case + ".grid_thw"], RG[case + ".input_ids"]
    feats = model.vision_tower(torch.from_numpy(pix), thw)
    e = _rel_rms_err(feats, torch.from_numpy(RG[p + "ref_image_features"]))
    assert e < 2e-2, e
    toks, lps = [], []
    for t, lp in generate_step(ids, model, torch.from_numpy(pix), None, max_tokens=8, temperature=0.0, image_grid_thw=thw):
        toks.append(t)
        lps.append(lp.float().cpu())
    ref_logits = torch.from_numpy(np.concatenate([RG[p + "ref_prefill_logits"][-1:], RG[p + "ref_decode_logits"]]))
    ok, n, margin = _tie_aware_equal(toks, RG[p + "ref_greedy"].tolist(), ref_logits, tol=3e-2)
    assert ok, (toks, RG[p + "ref_greedy"].tolist(), n, margin)
    ref_lp0 = O.logprobs_from_logits(ref_logits[0][None].to(BF))[0]
    ok, rep = bf16_close(lps[0], ref_lp0, ulps=2, atol_rms=3e-2)
    assert ok, rep


def test_reference_generate_step_text_golden_through_hip_path(tiny):
    """The reference's generate_step on a text prompt (tokens + bf16 logprobs of every step)."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    ids = RG["generate_step.text.input_ids"]
    toks, lps = [], []
    for t, lp in generate_step(ids, model, None, None, max_tokens=8, temperature=0.0):
        toks.append(t)
        lps.append(lp.float().cpu())
    ref_lp = torch.from_numpy(RG["generate_step.text.logprobs"])
    ok, n, margin = _tie_aware_equal(toks, RG["generate_step.text.tokens"].tolist(), ref_lp, tol=3e-2)
    assert ok, (toks, RG["generate_step.text.tokens"].tolist(), n, margin)
    for i in range(len(toks)):
        if toks[: i + 1] != RG["generate_step.text.tokens"].tolist()[: i + 1]:
            break
        ok, rep = bf16_close(lps[i], ref_lp[i].to(BF), ulps=2, atol_rms=3e-2)
        assert ok, (i, rep)


@pytest.mark.parametrize("B", [1, 2])
def test_paged_and_identity_kv_layouts_generate_the_same_tokens(B):
    """The identity layout (decode kernels compute page numbers) and the shared-free-list paged layout (kernels walk
    the block table) are two placements of the same cache: identical tokens and first-step logprobs."""
    from mlx_vlm_amd.generate import batch_generate_ids, generate_step

    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    outs = []
    for layout in ("identity", "paged"):
        model = build_product_model(cfg, W, kv_pool_tokens=4096, max_seqs=4, kv_layout=layout)
        assert model.language_model.pool.identity == (layout == "identity")
        if B == 1:
            ids, pix, thw = synth_request(cfg, [(56, 84)], n_text=14, seed=8)
            toks, lp0 = [], None
            for t, lp in generate_step(ids, model, torch.from_numpy(pix), None, max_tokens=70, temperature=0.0,
                                       image_grid_thw=thw):
                toks.append(t)
                lp0 = lp.float().cpu() if lp0 is None else lp0
            outs.append((toks, lp0))
        else:
            reqs = [synth_request(cfg, [(56, 84)], n_text=10 + 3 * i, seed=20 + i) for i in range(B)]
            toks, _ = batch_generate_ids(model, [r[0].reshape(-1) for r in reqs], [torch.from_numpy(r[1]) for r in reqs],
                                         [r[2] for r in reqs], max_tokens=40)
            outs.append(([list(t) for t in toks], None))
        del model
    assert outs[0][0] == outs[1][0]
    if outs[0][1] is not None:
        assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dims", ["2B", "7B"])
def test_real_width_two_layer_model_vs_oracle(dims):
    """Qwen2-VL-2B / -7B layer WIDTHS (hidden 1536 / 3584, GQA 12:2 / 28:4, inter 8960 / 18944, untied head for 7B,
    full-width ViT blocks with 16 heads of 80) at depth 2 and a small vocabulary: the shapes the kernels are tuned
    for (row-wave and split-K GEMVs, G = 6 / 7 decode attention, 256x256 GEMM on the ViT) against the oracle."""
    from mlx_vlm_amd.generate import generate_step

    if dims == "2B":
        text = oq.TextCfg(hidden_size=1536, num_hidden_layers=2, intermediate_size=8960, num_attention_heads=12,
                          num_key_value_heads=2, vocab_size=2048, tie_word_embeddings=True)
        hid = 1536
    else:
        text = oq.TextCfg(hidden_size=3584, num_hidden_layers=2, intermediate_size=18944, num_attention_heads=28,
                          num_key_value_heads=4, vocab_size=2048, tie_word_embeddings=False)
        hid = 3584
    cfg = oq.Cfg(text=text, vision=oq.VisionCfg(depth=2, embed_dim=1280, hidden_size=hid, num_heads=16),
                 image_token_id=2001, video_token_id=2002, vision_start_token_id=2003)
    W = oq.random_weights(cfg, seed=77, dtype=BF, std=0.02, embed_std=0.2)
    model = build_product_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    ids, pix, thw = synth_request(cfg, [(112, 168)], n_text=20, seed=5, text_hi=2000)
    feats = model.vision_tower(torch.from_numpy(pix), thw)
    ref_feats = oq.vision_tower(W, cfg, torch.from_numpy(pix).to(BF), thw)
    assert _rel_rms_err(feats, ref_feats) < 2e-2
    n_new = 12
    ref_toks, ref_logits = oq.generate_greedy(W, cfg, ids, torch.from_numpy(pix).to(BF), thw, max_tokens=n_new,
                                              return_logits=True)
    toks, lps = [], []
    for t, lp in generate_step(ids, model, torch.from_numpy(pix), None, max_tokens=n_new, temperature=0.0, image_grid_thw=thw):
        toks.append(t)
        lps.append(lp.float().cpu())
    ok, n, margin = _tie_aware_equal(toks, ref_toks, ref_logits, tol=3e-2)
    assert ok, (toks, ref_toks, n, margin)
    ok, rep = bf16_close(lps[0], O.logprobs_from_logits(ref_logits[0][None])[0], ulps=2, atol_rms=3e-2)
    assert ok, rep


def _write_tiny_checkpoint(tmp_path, cfg, W):
    """HF Qwen2-VL layout on disk: config.json (text parameters at the root), model.safetensors with the classic HF
    key names (visual.* / model.* ; conv weight (O,C,T,H,W)), a WordLevel tokenizer with the special vision tokens."""
    import json

    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    t, v = cfg.text, cfg.vision
    conf = dict(model_type="qwen2_vl", hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
                intermediate_size=t.intermediate_size, num_attention_heads=t.num_attention_heads,
                num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps,
                rope_theta=t.rope_theta, rope_scaling={"type": "mrope", "mrope_section": list(t.mrope_section)},
                tie_word_embeddings=t.tie_word_embeddings, image_token_id=cfg.image_token_id,
                video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_token_id, eos_token_id=[1],
                vision_config=dict(model_type="qwen2_vl", depth=v.depth, embed_dim=v.embed_dim, hidden_size=v.hidden_size,
                                   num_heads=v.num_heads, patch_size=v.patch_size, mlp_ratio=int(v.mlp_ratio),
                                   in_channels=v.in_channels, spatial_merge_size=v.spatial_merge_size,
                                   temporal_patch_size=v.temporal_patch_size))
    (tmp_path / "config.json").write_text(json.dumps(conf))
    hf = {}
    for k, w in W.items():
        w = w.contiguous()
        if k.startswith("vision_tower."):
            hk = "visual." + k[len("vision_tower."):]
            if "patch_embed.proj.weight" in k:
                w = w.permute(0, 4, 1, 2, 3).contiguous()          # (O,T,H,W,C) -> HF (O,C,T,H,W)
        elif k.startswith("language_model.model."):
            hk = "model." + k[len("language_model.model."):]
        else:
            hk = "lm_head." + k[len("language_model.lm_head."):]
        hf[hk] = w
    save_file(hf, str(tmp_path / "model.safetensors"))
    vocab = {f"t{i}": i for i in range(t.vocab_size)}
    for name, idx in (("<unk>", 0), ("<eos>", 1), ("<pad>", 2)):      # specials match as substrings: keep them off the "tN" words
        del vocab[f"t{idx}"]
        vocab[name] = idx
    for name, idx in (("<|image_pad|>", cfg.image_token_id), ("<|video_pad|>", cfg.video_token_id),
                      ("<|vision_start|>", cfg.vision_start_token_id), ("<|vision_end|>", cfg.vision_start_token_id + 1)):
        del vocab[f"t{idx}"]
        vocab[name] = idx
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<eos>", pad_token="<pad>", unk_token="<unk>")
    # special (added) tokens are split out before pre-tokenisation: the expanded "<|image_pad|><|image_pad|>..." run has no spaces
    fast.add_special_tokens({"additional_special_tokens": ["<|image_pad|>", "<|video_pad|>", "<|vision_start|>", "<|vision_end|>"]})
    fast.save_pretrained(str(tmp_path))
    (tmp_path / "preprocessor_config.json").write_text(json.dumps(dict(image_mean=[0.5] * 3, image_std=[0.5] * 3)))


def test_load_and_generate_user_api_end_to_end(tmp_path):
    """load(path) -> (model, processor); generate(model, processor, prompt, image=...) through tokenizer, image processor,
    placeholder expansion, ViT, prefill, graph decode, detokenizer - against the oracle on the same checkpoint."""
    from mlx_vlm_amd import generate, load, stream_generate

    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=4321, dtype=BF, std=0.05, embed_std=0.2)
    _write_tiny_checkpoint(tmp_path, cfg, W)
    model, processor = load(str(tmp_path), kv_pool_tokens=4096, max_seqs=4)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (100, 150, 3), dtype=np.uint8)          # not a multiple of 28: exercises smart resize
    words = " ".join(f"t{int(i)}" for i in rng.integers(3, 1000, 9))
    prompt = f"<|vision_start|> <|image_pad|> <|vision_end|> {words}"
    out = generate(model, processor, prompt, image=img, max_tokens=10, temperature=0.0)
    # the oracle on the same inputs
    from oracle import image_processor as oip
    pix, thw = oip.process([img.transpose(2, 0, 1)])
    n_img = int(thw.prod()) // 4
    ids = [cfg.vision_start_token_id] + [cfg.image_token_id] * n_img + [cfg.vision_start_token_id + 1] + \
          [int(w[1:]) for w in words.split()]
    ref_toks, ref_logits = oq.generate_greedy(W, cfg, np.array([ids]), torch.from_numpy(pix).to(BF), thw, max_tokens=10,
                                              return_logits=True)
    assert out.prompt_tokens == len(ids)
    chunks = list(stream_generate(model, processor, prompt, image=img, max_tokens=10, temperature=0.0))
    assert chunks[-1].finish_reason is not None          # closing chunk re-yields the last token with the flushed text
    got = [r.token for r in chunks[:-1]]
    if 1 in ref_toks:                      # <eos> stops generation like the reference's StoppingCriteria
        ref_toks = ref_toks[:ref_toks.index(1)]
    ok, n, margin = _tie_aware_equal(got[:len(ref_toks)], ref_toks, ref_logits, tol=3e-2)
    assert ok, (got, ref_toks, n, margin)
    # streamed text segments concatenate to the tokenizer's own decode of the generated ids (reference tokenizer_utils.py:19-86)
    segs = "".join(r.text for r in chunks)
    assert segs.split() == processor.tokenizer.decode(got).split() and out.text == segs
    assert out.finish_reason in ("stop", "length") and out.generation_tokens >= 1 and out.prompt_tps > 0
    # batch_generate (continuous batching underneath): image + text-only + image requests, same tokens as the stream
    from mlx_vlm_amd import BatchGenerator, batch_generate
    resp = batch_generate(model, processor, images=[img, None, img], prompts=[prompt, words, prompt], max_tokens=6)
    assert len(resp.texts) == 3 and resp.texts[0] == resp.texts[2] and resp.stats.generation_tokens >= 3
    assert len(resp.tokens[0]) >= 1 and resp.tokens[0] == got[:len(resp.tokens[0])]
    static = batch_generate(model, processor, images=[img, None, img], prompts=[prompt, words, prompt], max_tokens=6,
                            continuous=False)
    assert static.tokens == resp.tokens
    assert resp.image_sizes == [(100, 150), (0, 0), (100, 150)]          # original (height, width), (0, 0) without an image
    assert batch_generate(model, processor, images=[img], prompts=[prompt], max_tokens=2, group_by_shape=False,
                          track_image_sizes=False).image_sizes is None
    assert callable(BatchGenerator)


# ---- continuous batching: BatchGenerator over the paged pool
def _mixed_requests(cfg, n, seed0=40):
    reqs = []
    for i in range(n):
        if i % 3 == 2:
            reqs.append((np.random.default_rng(seed0 + i).integers(3, 1000, (1, 6 + 2 * i)), None, None))
        else:
            reqs.append(synth_request(cfg, [(56, 56 + 28 * (i % 2))], n_text=5 + i, seed=seed0 + i))
    return reqs


def _single_runs(model, reqs, max_tokens):
    from mlx_vlm_amd.generate import generate_step

    outs = []
    for (ids, pix, thw), m in zip(reqs, max_tokens):
        kw = dict(image_grid_thw=thw) if thw is not None else {}
        outs.append([(t, float(lp[t])) for t, lp in generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None,
                                                                 None, max_tokens=m, **kw)])
    return outs


def _insert_all(gen, reqs, max_tokens):
    kw = [dict(pixel_values=torch.from_numpy(p), image_grid_thw=g) if p is not None else {} for _, p, g in reqs]
    return gen.insert([r[0].reshape(-1) for r in reqs], list(max_tokens), prompt_kwargs=kw)


# A request decoded inside a batch and the same request decoded alone run DIFFERENT reduction structures since round 3 (one
# row: page-split attention merged in the o_proj prologue; 2..8 rows: merged by the attention launch's last
# arriver from fp32 partials; 9..16 rows: projections on the matrix cores): every step's hidden state may differ by a bf16 ulp.
# A token log-prob is bf16(logit - bf16(logsumexp)): one ulp of a logsumexp in [8, 16) is 0.0625, the logit adds its own, so
# two correct runs agree to LP_ATOL; greedy tokens agree until a step whose top two candidates lie inside that noise.
LP_ATOL = 0.13


def _assert_streams_equal_up_to_ties(got, singles, min_equal=0.9):
    """got / singles: per request [(token, token log-prob)].  Every compared step's log-probs within LP_ATOL; tokens equal up
    to the first tie (after which the two streams feed different tokens and are no longer comparable); at least `min_equal`
    of all tokens compared equal."""
    n_equal = n_total = 0
    for u, (a, b) in enumerate(zip(got, singles)):
        assert len(a) == len(b), (u, a, b)
        n_total += len(b)
        for i, ((ta, la), (tb, lb)) in enumerate(zip(a, b)):
            assert abs(la - lb) <= LP_ATOL, (u, i, la, lb, a, b)
            if ta != tb:
                break
            n_equal += 1
    assert n_equal >= min_equal * n_total, (n_equal, n_total)


@pytest.mark.parametrize("async_prefill", [True, False])
def test_batch_generator_continuous_equals_single_requests(tiny, async_prefill):
    """11 requests (images and text, different lengths, different max_tokens) through 4 decode rows: rows finish at
    different steps, the last row moves into the hole, queued prompts are admitted as rows free up, the step width
    goes 4 -> 2 -> 1 at the end.  Every request must produce the tokens (and the token logprobs) it produces alone - up
    to bf16 ties, see LP_ATOL; the reference's finish rules (ar.py:1313-1316) and response fields are checked on the way.
    async_prefill: admissions run on a second stream under the decode steps and join when their event fires (the
    round at which a request joins then depends on timing - its tokens must not)."""
    from mlx_vlm_amd.batch import BatchGenerator

    cfg, W, model = tiny
    reqs = _mixed_requests(cfg, 11)
    max_tokens = [4 + (5 * i) % 11 for i in range(11)]
    singles = _single_runs(model, reqs, max_tokens)
    free_before = len(model.language_model.pool._free_seqs)
    gen = BatchGenerator(model, None, max_tokens=7, completion_batch_size=4, prefill_batch_size=2,
                         async_prefill=async_prefill)
    uids = _insert_all(gen, reqs, max_tokens)
    assert uids == list(range(11)) and gen.has_pending_prompts and len(gen.unprocessed_prompts) == 11
    got = {u: [] for u in uids}
    finished, prompt_seen, widths = {}, {}, set()
    rounds = 0
    while gen.has_work:
        prompts, out = gen.next()
        rounds += 1
        assert len(gen) <= 4 and rounds < 400
        widths.add(gen._width)
        for p in prompts:
            prompt_seen[p.uid] = p.prompt_tokens
        seen = set()
        for r in out:
            assert r.uid not in finished and r.uid not in seen          # one token per running request per round
            seen.add(r.uid)
            got[r.uid].append((r.token, r.token_logprob))
            if r.finish_reason is not None:
                finished[r.uid] = r.finish_reason
    assert widths >= {1, 2, 4}
    assert prompt_seen == {u: reqs[u][0].size for u in uids}
    assert finished == {u: "length" for u in uids}
    _assert_streams_equal_up_to_ties([got[u] for u in uids], singles)
    st = gen.stats()
    assert st.generation_tokens == sum(max_tokens) and st.prompt_tokens == sum(r[0].size for r in reqs)
    assert st.prompt_tps > 0 and st.generation_tps > 0
    gen.close()
    assert len(model.language_model.pool._free_seqs) == free_before      # every sequence and the scratch page returned


@pytest.mark.parametrize("async_prefill", [True, False])
def test_batch_generator_per_request_logits_processors_equal_single_requests(tiny, async_prefill):
    """BatchGenerator.insert(..., logits_processors=) (reference ar.py:2584-2606): 10 requests through 4 decode rows, each
    with ITS OWN processors - repetition penalty, presence + frequency penalties with short contexts, a logit_bias list,
    all of them, or none - must produce exactly the tokens and token log-probs they produce alone through generate_step
    with the same keywords (the single-stream device pass, pinned to the reference's processors in test_ops_gpu.py).
    Rows finish at different steps and move (their history / parameter rows move with them); a row that had processors
    is re-used by a request without any."""
    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.generate import generate_step
    from mlx_vlm_amd.sample_utils import make_logits_processors

    cfg, W, model = tiny
    reqs = _mixed_requests(cfg, 10, seed0=240)
    max_tokens = [6 + (5 * i) % 9 for i in range(10)]
    kws = [dict(repetition_penalty=1.3, repetition_context_size=12), None, dict(presence_penalty=0.75, presence_context_size=5,
           frequency_penalty=0.25, frequency_context_size=9), dict(logit_bias={7: 4.0, 901: -3.0, 333: 2.5}), None,
           dict(repetition_penalty=1.1, presence_penalty=0.5, frequency_penalty=0.125, logit_bias={5: 1.0}),
           dict(repetition_penalty=1.5), None, dict(frequency_penalty=0.5, frequency_context_size=20), dict(logit_bias={11: 6.0})]
    singles = []
    for (ids, pix, thw), m, kw in zip(reqs, max_tokens, kws):
        a = dict(image_grid_thw=thw) if thw is not None else {}
        a.update(kw or {})
        singles.append([(t, float(lp[t])) for t, lp in generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None,
                                                                      None, max_tokens=m, **a)])
    plain = _single_runs(model, reqs, max_tokens)
    assert sum(a != b for a, b in zip(singles, plain)) >= 4              # the processors do change these streams
    gen = BatchGenerator(model, None, max_tokens=7, completion_batch_size=4, prefill_batch_size=2, async_prefill=async_prefill)
    pk = [dict(pixel_values=torch.from_numpy(p), image_grid_thw=g) if p is not None else {} for _, p, g in reqs]
    specs = [make_logits_processors(**kw) if kw else None for kw in kws]
    uids = gen.insert([r[0].reshape(-1) for r in reqs], list(max_tokens), prompt_kwargs=pk, logits_processors=specs)
    got = {u: [] for u in uids}
    while gen.has_work:
        _, out = gen.next()
        for r in out:
            got[r.uid].append((r.token, r.token_logprob))
    gen.close()
    _assert_streams_equal_up_to_ties([got[u] for u in uids], singles)
    # the processors really reached the rows: the batch's streams follow the processed singles, not the plain ones
    assert sum([t for t, _ in got[u]] != [t for t, _ in plain[u]] for u in uids) >= 2


def _torch_repetition_penalty(penalty, context_size):
    """The reference's repetition penalty (sample_utils.py:424-447) written as a Python callable over DEVICE tensors - what a
    drop-in caller would pass as `logits_processors=[...]`"""
    def proc(tokens, logits):
        assert tokens.dtype == torch.int32 and tokens.is_cuda and logits.is_cuda and logits.shape[0] == 1
        idx = tokens[-context_size:].long().unique()
        sel = logits[:, idx].float()
        sel = torch.where(sel < 0, (sel * penalty).to(torch.bfloat16).float(), (sel / penalty).to(torch.bfloat16).float())
        out = logits.clone()
        out[:, idx] = sel.to(out.dtype)
        return out
    return proc


@pytest.mark.parametrize("sizes", [[(56, 84)], []])
def test_generate_step_python_callables_take_the_eager_step(tiny, sizes):
    """generate_step(sampler=<callable>, logits_processors=[<callables>]) (reference ar.py:151-193,360-379): the eager step.
    (1) identity processor + argmax sampler == the captured step, tokens and log-probs bit for bit (same kernels, the host
    in the middle); (2) a repetition penalty written as a Python callable over the device tensors == the built-in device
    pass with the same parameters (tokens identical, log-probs 2 ulps): `tokens` really is prompt + every fed token."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    ids, pix, thw = synth_request(cfg, sizes, n_text=14, seed=33) if sizes else \
        (np.random.default_rng(34).integers(3, 1000, (1, 19)), None, None)
    kw = dict(image_grid_thw=thw) if thw is not None else {}
    pv = (lambda: torch.from_numpy(pix)) if pix is not None else (lambda: None)

    def run(**a):
        toks, lps = [], []
        for t, lp in generate_step(ids, model, pv(), None, max_tokens=24, temperature=0.0, **kw, **a):
            toks.append(t)
            lps.append(lp.float().cpu())
        return toks, torch.stack(lps)

    base_t, base_lp = run()
    seen = []

    def ident(tokens, logits):
        seen.append(int(tokens.numel()))
        return logits

    t, lp = run(sampler=lambda logprobs: logprobs.argmax(-1), logits_processors=[ident])
    assert t == base_t and torch.equal(lp, base_lp)
    assert seen == [ids.size + i for i in range(24)]                       # the token context grows by one fed token per step
    rep_t, rep_lp = run(repetition_penalty=1.4, repetition_context_size=16)
    t, lp = run(logits_processors=[_torch_repetition_penalty(1.4, 16)])
    assert rep_t != base_t and t == rep_t
    ok, msg = bf16_close(lp, rep_lp, ulps=2, atol_rms=5e-3)
    assert ok, msg
    # a spec and a callable together: spec first (device pass), then the callable
    t2, _ = run(logits_processors=[__import__("mlx_vlm_amd").sample_utils.make_logits_processors(logit_bias={7: 3.0}), ident])
    t3, _ = run(logit_bias={7: 3.0})
    assert t2 == t3


def test_generate_step_thinking_budget_forces_the_closing_sequence(tiny):
    """generate_step(thinking_budget_criteria=) (ar.py:308,510-513) with the caller driving the criteria as stream_generate
    does (dispatch.py:1016-1018): past the budget the next tokens are "\\n" + the end marker, then decoding continues from
    them; before that the stream equals the plain one."""
    from mlx_vlm_amd.generate import generate_step
    from mlx_vlm_amd.utils import ThinkingBudgetCriteria

    cfg, W, model = tiny

    class Tok:
        def encode(self, t, add_special_tokens=False):
            return {"<think>": [990], "</think>": [991], "\n": [992]}[t]

    ids = np.random.default_rng(35).integers(3, 900, (1, 17))
    plain = [t for t, _ in generate_step(ids, model, None, None, max_tokens=12, temperature=0.0)]
    crit = ThinkingBudgetCriteria(Tok(), thinking_budget=4, thinking_start_token="<think>", enable_thinking=True,
                                  prompt_preopens_thinking=True)
    got = []
    for t, _ in generate_step(ids, model, None, None, max_tokens=12, temperature=0.0, thinking_budget_criteria=crit):
        got.append(t)
        crit(t)
    assert got[:5] == plain[:5] and got[5:7] == [992, 991] and len(got) == 12
    # the forced tokens were FED: the continuation equals a plain run over prompt + the tokens so far
    cont = [t for t, _ in generate_step(np.concatenate([ids, np.asarray([got[:7]])], axis=1), model, None, None, max_tokens=5,
                                        temperature=0.0)]
    assert got[7:12] == cont


@pytest.mark.parametrize("py_sampler", [False, True])
def test_batch_generator_python_callables_equal_single_requests(tiny, py_sampler):
    """BatchGenerator.insert(..., logits_processors=[[callable]], thinking_budget_criteria=[...]) and
    BatchGenerator(sampler=<callable>) (reference ar.py:1044-1141,1303-1350,2584-2606): 7 requests through 4 rows, some with
    a Python repetition penalty, one with a thinking budget, the rest plain - eager steps while a callable's row is live,
    captured steps otherwise.  Each request's tokens equal the ones it produces alone through generate_step with the same
    callables (up to bf16 ties: a batched step and a one-row step reduce in different orders)."""
    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.generate import generate_step
    from mlx_vlm_amd.utils import ThinkingBudgetCriteria

    cfg, W, model = tiny

    class Tok:
        def encode(self, t, add_special_tokens=False):
            return {"<think>": [990], "</think>": [991], "\n": [992]}[t]

    def budget():
        return ThinkingBudgetCriteria(Tok(), thinking_budget=3, thinking_start_token="<think>", enable_thinking=True,
                                      prompt_preopens_thinking=True)

    reqs = _mixed_requests(cfg, 7, seed0=340)
    max_tokens = [9, 12, 7, 11, 8, 10, 6]
    procs = [[_torch_repetition_penalty(1.3, 12)], None, None, [_torch_repetition_penalty(1.5, 20)], None, None, None]
    crits = [None, None, budget(), None, None, None, None]
    smp = (lambda logprobs: logprobs.float().argmax(-1)) if py_sampler else None
    singles = []
    for (ids, pix, thw), m, pr, has_budget in zip(reqs, max_tokens, procs, [c is not None for c in crits]):
        a = dict(image_grid_thw=thw) if thw is not None else {}
        c = budget() if has_budget else None
        row = []
        for t, lp in generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None, None, max_tokens=m,
                                   logits_processors=pr, sampler=smp, thinking_budget_criteria=c, **a):
            row.append((t, float(lp[t])))
            if c is not None:
                c(t)
        singles.append(row)
    assert [t for t, _ in singles[2]][4:6] == [992, 991]
    gen = BatchGenerator(model, None, max_tokens=7, completion_batch_size=4, prefill_batch_size=2, sampler=smp)
    pk = [dict(pixel_values=torch.from_numpy(p), image_grid_thw=g) if p is not None else {} for _, p, g in reqs]
    uids = gen.insert([r[0].reshape(-1) for r in reqs], list(max_tokens), prompt_kwargs=pk, logits_processors=procs,
                      thinking_budget_criteria=crits)
    got = {u: [] for u in uids}
    while gen.has_work:
        _, out = gen.next()
        for r in out:
            got[r.uid].append((r.token, r.token_logprob))
    gen.close()
    # (the forced tokens report the log-prob of the token the model had sampled, as the reference does: compare tokens there)
    for u, ref in zip(uids, singles):
        if u == uids[2]:
            assert [t for t, _ in got[u]] == [t for t, _ in ref]
    _assert_streams_equal_up_to_ties([got[u] for i, u in enumerate(uids) if i != 2], [s for i, s in enumerate(singles) if i != 2])


def test_batch_generator_16_rows_matrix_core_steps_equal_single_requests(tiny):
    """20 requests through 16 decode rows: steps of 16 and 8 rows run their projections on the matrix cores
    (csrc/gemv_mfma.hip), narrower ones on the v_dot2c GEMVs, single requests on the latter only.  The two kernel families
    sum in different orders, so a request's tokens must equal the ones it produces alone except where its own top-2
    log-prob gap is inside bf16 noise, and the token log-probs agree to 2 bf16 ulps."""
    from mlx_vlm_amd.batch import BatchGenerator

    cfg, W, _ = tiny
    model = build_product_model(cfg, W, kv_pool_tokens=16384, max_seqs=40)       # 16 rows + admissions ahead + scratch
    reqs = _mixed_requests(cfg, 20, seed0=140)
    max_tokens = [6 + (7 * i) % 9 for i in range(20)]
    singles = _single_runs(model, reqs, max_tokens)
    gen = BatchGenerator(model, None, max_tokens=8, completion_batch_size=16, prefill_batch_size=8)
    assert gen.completion_batch_size == 16
    uids = _insert_all(gen, reqs, max_tokens)
    got = {u: [] for u in uids}
    widths = set()
    while gen.has_work:
        _, out = gen.next()
        widths.add(gen._width)
        for r in out:
            got[r.uid].append((r.token, r.token_logprob))
    gen.close()
    assert 16 in widths and 8 in widths
    assert all(len(got[u]) == max_tokens[u] for u in uids)
    _assert_streams_equal_up_to_ties([got[u] for u in uids], singles)


@pytest.mark.parametrize("rows", [32, 64])
def test_batch_generator_wide_steps_on_the_prefill_gemms_equal_single_requests(tiny, rows):
    """More than 16 decode rows on request (completion_batch_size = 32 / 64): such WIDE steps run a layer as the prefill's
    launch sequence - RMSNorm, GEMMs, M-RoPE + KV write at the rows' device-resident slots - with the paged decode attention
    (engine.hip decode_impl); captured in the step's graph with the launch stream's split-K workspace.  Every request must
    produce the tokens it produces alone up to its own bf16 ties (another kernel family = another summation order)."""
    from mlx_vlm_amd.batch import BatchGenerator

    cfg, W, _ = tiny
    n = rows + 6
    model = build_product_model(cfg, W, kv_pool_tokens=32768, max_seqs=2 * rows + 8)
    reqs = _mixed_requests(cfg, n, seed0=400 + rows)
    max_tokens = [5 + (5 * i) % 8 for i in range(n)]
    singles = _single_runs(model, reqs, max_tokens)
    gen = BatchGenerator(model, None, max_tokens=8, completion_batch_size=rows, prefill_batch_size=rows)
    assert gen.completion_batch_size == rows
    uids = _insert_all(gen, reqs, max_tokens)
    got = {u: [] for u in uids}
    widths = set()
    while gen.has_work:
        _, out = gen.next()
        widths.add(gen._width)
        for r in out:
            got[r.uid].append((r.token, r.token_logprob))
    gen.close()
    assert rows in widths
    assert all(len(got[u]) == max_tokens[u] for u in uids)
    _assert_streams_equal_up_to_ties([got[u] for u in uids], singles)


def test_batch_generator_wide_rows_are_refused_where_the_engine_has_none(tiny):
    """a language model that caps its decode rows (phi3_v: two-table RoPE) keeps 16-row steps whatever is asked for"""
    from mlx_vlm_amd.batch import BatchGenerator

    cfg, W, _ = tiny
    model = build_product_model(cfg, W, kv_pool_tokens=16384, max_seqs=80)
    model.language_model.MAX_DECODE_ROWS = 16
    gen = BatchGenerator(model, None, completion_batch_size=32)
    assert gen.completion_batch_size == 16
    gen.close()
    del model.language_model.MAX_DECODE_ROWS


def test_batch_generator_stop_token_and_remove(tiny):
    """A stop token ends one request with finish_reason "stop" (the token is still reported, as in the reference);
    remove(uid) drops a running request between rounds; the other rows are untouched by either."""
    from mlx_vlm_amd.batch import BatchGenerator

    cfg, W, model = tiny
    reqs = _mixed_requests(cfg, 5, seed0=70)
    singles = [[t for t, _ in s] for s in _single_runs(model, reqs, [12] * 5)]
    stop_tok = singles[1][4]
    want = []
    for s in singles:                       # what each request emits with that stop token in force
        want.append(s[:s.index(stop_tok) + 1] if stop_tok in s else s)
    gen = BatchGenerator(model, None, max_tokens=12, stop_tokens={stop_tok}, completion_batch_size=8, compute_logprobs=False)
    uids = _insert_all(gen, reqs, [12] * 5)
    victim = next(u for u in uids if u != 1 and len(want[u]) >= 6)
    got = {u: [] for u in uids}
    reasons = {}
    removed = False
    while gen.has_work:
        _, out = gen.next()
        for r in out:
            got[r.uid].append(r.token)
            assert r.token_logprob == 0.0
            if r.finish_reason:
                reasons[r.uid] = r.finish_reason
        if len(got[victim]) == 3 and not removed:
            removed = gen.remove(victim)
            assert removed and not gen.remove(12345)
    assert removed and len(got[victim]) == 3 and victim not in reasons
    # The stop / remove semantics are checked on every stream ITSELF; agreement with the single-request runs is asked up to the
    # first near-tie (a request inside a batch and alone run different reduction structures, see LP_ATOL above: after a step
    # whose top two candidates lie inside that noise the two greedy streams feed different tokens).  Round 5's attention kernel
    # takes its row sums from the bf16 P it multiplies with and moved request 3's fifth step across such a tie.
    n_equal = n_total = 0
    for u in uids:
        g, w = got[u], want[u][:3] if u == victim else want[u]
        if u != victim:
            if stop_tok in g:
                assert g.index(stop_tok) == len(g) - 1 and reasons[u] == "stop", (u, g, reasons[u])
            else:
                assert len(g) == 12 and reasons[u] == "length", (u, g, reasons[u])
        k = 0
        while k < min(len(g), len(w)) and g[k] == w[k]:
            k += 1
        n_equal += k
        n_total += len(w)
        if k == len(w):
            assert len(g) == len(w), (u, g, w)          # same tokens: same end, same reason
    assert n_equal >= 0.8 * n_total, (n_equal, n_total, got, want)
    assert got[uids[1]][:5] != singles[1][:5] or reasons[uids[1]] == "stop"
    gen.close()


def test_generate_batch_continuous_more_requests_than_rows(tiny):
    from mlx_vlm_amd.batch import generate_batch_continuous

    cfg, W, model = tiny
    reqs = _mixed_requests(cfg, 13, seed0=90)
    singles = [[t for t, _ in s] for s in _single_runs(model, reqs, [9] * 13)]
    toks, stats = generate_batch_continuous(model, [r[0].reshape(-1) for r in reqs],
                                            [torch.from_numpy(r[1]) if r[1] is not None else None for r in reqs],
                                            [r[2] for r in reqs], max_tokens=9)
    assert toks == singles
    assert stats.generation_tokens == 9 * 13 and stats.generation_tps > 0


# ------------------------------------------------------------------------------------------------ MLX 4-bit checkpoints
def _quantized_tiny(seed=1234):
    """tiny Qwen2-VL with the language model quantized the way `mlx_vlm.convert -q` leaves it (vision tower skipped):
    -> (cfg, checkpoint-style dict with .weight/.scales/.biases, oracle dict with QW weights)"""
    from oracle import quant as Q

    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=seed, dtype=BF, std=0.05, embed_std=0.2)
    ck, ow = Q.quantize_checkpoint(W, predicate=lambda p, v: p.startswith("language_model."))
    return cfg, ck, ow


def test_quantized_4bit_teacher_forced_decode_vs_oracle():
    """4-bit language model (embed_tokens, every Linear, the tied head): prefill through dequant + bf16 GEMM, decode through
    the fused 4-bit GEMVs, every teacher-forced step's logits against the oracle running the SAME quantized weights through
    nn.QuantizedLinear / nn.QuantizedEmbedding restated (oracle/quant.py).  Tolerance as the bf16 tiny test (2e-2 rel-rms
    per row); prefill multiplies bf16-rounded dequantized weights (the qmm form), decode the fp32 affine form (qmv)."""
    cfg, ck, ow = _quantized_tiny()
    model = build_product_model(cfg, ck, kv_pool_tokens=4096, max_seqs=4)
    lm = model.language_model
    assert lm.quantized
    ids, pix, thw = synth_request(cfg, [(56, 84)], n_text=14, seed=44)
    forced = np.random.default_rng(45).integers(3, 1000, 70)
    ref = oq.decode_teacher_forced(ow, cfg, ids, torch.from_numpy(pix).to(BF), thw, forced)
    f = model.get_input_embeddings(ids, torch.from_numpy(pix), image_grid_thw=thw)
    cache = lm.make_cache()
    out = lm(ids, f.inputs_embeds, cache=cache, position_ids=f.position_ids, rope_deltas=f.rope_deltas, logits_to_keep=1)
    rows = [out.logits[0, -1].clone()]
    for y in forced:
        rows.append(lm(np.array([[int(y)]]), cache=cache).logits[0, -1].clone())
    cache[0]._seq.release()
    got = torch.stack(rows)
    worst = 0.0
    for i in range(ref.shape[0]):
        e = _rel_rms_err(got[i], ref[i])
        worst = max(worst, e)
        assert e < 2e-2, (i, e)
    ok, rep = bf16_close(got, ref, ulps=4, atol_rms=8e-2)
    assert ok, rep
    # the quantization itself is visible: the bf16 model's logits are far from these (the test is not vacuous)
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    ref_bf = oq.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pix).to(BF), thw, forced[:4])
    assert _rel_rms_err(ref[:5], ref_bf) > 0.1
    print(f"4-bit teacher-forced: worst row rel-rms {worst:.4f}")


def test_quantized_4bit_checkpoint_load_and_generate(tmp_path):
    """An MLX-format 4-bit checkpoint on disk (uint32 `weight`, `scales`, `biases`, config["quantization"]) through load()
    -> generate_step with the captured decode graph, B = 1 and a batch of 2, against the oracle on the same weights."""
    import json

    from mlx_vlm_amd import load
    from mlx_vlm_amd.generate import generate_step

    cfg, ck, ow = _quantized_tiny(seed=4321)
    ck = {k: (v.view(torch.uint32) if v.dtype == torch.int32 else v) for k, v in ck.items()}
    _write_tiny_checkpoint(tmp_path, cfg, ck)
    conf = json.loads((tmp_path / "config.json").read_text())
    conf["quantization"] = {"group_size": 64, "bits": 4}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    model, _ = load(str(tmp_path), kv_pool_tokens=4096, max_seqs=4)
    assert model.language_model.quantized
    ids, pix, thw = synth_request(cfg, [(56, 56)], n_text=11, seed=46)
    n_new = 12
    ref_toks, ref_logits = oq.generate_greedy(ow, cfg, ids, torch.from_numpy(pix).to(BF), thw, max_tokens=n_new,
                                              return_logits=True)
    toks, lps = [], []
    for t, lp in generate_step(ids, model, torch.from_numpy(pix), None, max_tokens=n_new, temperature=0.0, image_grid_thw=thw):
        toks.append(t)
        lps.append(lp.float().cpu())
    ok, n, margin = _tie_aware_equal(toks, ref_toks, ref_logits, tol=3e-2)
    assert ok, (toks, ref_toks, n, margin)
    ok, rep = bf16_close(lps[0], O.logprobs_from_logits(ref_logits[0][None])[0], ulps=2, atol_rms=3e-2)
    assert ok, rep
    # unsupported modes fail loudly
    conf["quantization"] = {"group_size": 32, "bits": 8}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    with pytest.raises(NotImplementedError):
        load(str(tmp_path))


def test_quantized_4bit_16_row_batch_equals_single_requests():
    """A 4-bit language model through 16 decode rows (dequant-fused MFMA projections) vs the same requests alone (4-bit
    v_dot2c GEMVs): tokens equal except at ties inside bf16 noise; a token's log-prob is bf16(logit - bf16(lse)), so the two
    summation orders may differ by an ulp of the LOGIT: the bound is 2 ulps of the step's log-prob span (max |lp| over the
    vocabulary, the logit scale), not of the log-prob itself."""
    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.generate import generate_step

    cfg, ck, ow = _quantized_tiny()
    model = build_product_model(cfg, ck, kv_pool_tokens=16384, max_seqs=40)
    reqs = _mixed_requests(cfg, 18, seed0=240)
    max_tokens = [5 + (3 * i) % 7 for i in range(18)]
    singles = []
    for (ids, pix, thw), m in zip(reqs, max_tokens):
        kw = dict(image_grid_thw=thw) if thw is not None else {}
        singles.append([(t, float(lp[t]), float(lp.float().abs().max())) for t, lp in
                        generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None, None, max_tokens=m, **kw)])
    gen = BatchGenerator(model, None, max_tokens=8, completion_batch_size=16, prefill_batch_size=8)
    assert gen.completion_batch_size == 16
    uids = _insert_all(gen, reqs, max_tokens)
    got = {u: [] for u in uids}
    widths = set()
    while gen.has_work:
        _, out = gen.next()
        widths.add(gen._width)
        for r in out:
            got[r.uid].append((r.token, r.token_logprob))
    gen.close()
    assert 16 in widths
    n_equal = 0
    for u in uids:
        for i, ((ta, la), (tb, lb, span)) in enumerate(zip(got[u], singles[u])):
            assert abs(la - lb) <= 2 * 2 ** -7 * span, (u, i, span, got[u], singles[u])
            if ta != tb:
                break
            n_equal += 1
        assert len(got[u]) == len(singles[u]) == max_tokens[u]
    assert n_equal >= 0.8 * sum(max_tokens), (n_equal, sum(max_tokens))       # ties are the exception, not the rule
