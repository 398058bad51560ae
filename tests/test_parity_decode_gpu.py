"""GPU parity tests that can fail: the decode path checked step by step and at the sizes the
benchmark runs.

  * teacher-forced decode: a SEEDED RANDOM token stream (not the model's own argmax, which on random weights is a
    fixed point) is fed through the engine's decode forward and EVERY step's logits are compared with the oracle's
    (generate/ar.py:334-389 with the fed token prescribed; positions = cache offset + rope delta,
    language.py:476-509);
  * peaked head (SURVEY.md par. 8d): an untied lm_head built so that the next token is a permutation successor with a
    margin of > 0.5 rms at every step - greedy tokens through the captured decode graph must be IDENTICAL to the
    oracle's, no tie rule, and every step's bf16 log-probs are compared;
  * BASELINE configs[1] at full size: Qwen2-VL-2B dims, 32 ViT blocks, 28 layers, V = 151,936, one 448 x 448 image +
    128 text tokens: image features, last-row prefill logits and 8 teacher-forced decode steps against the oracle;
  * BASELINE configs[0] at full size: nanoLLaVA dims (SigLIP-so400m tower, Qwen1.5-0.5B), one 384 x 384 image.

Tolerances are stated next to each assert; a bf16 tensor after n layers differs from the oracle by the fp32
accumulation order of every GEMM (1-ulp flips that compound), measured here at 0.3-1.2 % of the tensor's rms.
"""
import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import qwen2_vl as oq
from tests.helpers import bf16_close, build_product_model, synth_request

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def _engine_teacher_forced(model, ids, pix, thw, forced):
    """prefill (last row) + one module-contract decode call per forced token -> logits [1 + len(forced), V] (device)"""
    lm = model.language_model
    kw = dict(image_grid_thw=thw) if thw is not None else {}
    f = model.get_input_embeddings(ids, torch.from_numpy(pix) if pix is not None else None, **kw)
    cache = lm.make_cache()
    out = lm(ids, f.inputs_embeds, cache=cache, position_ids=f.position_ids, rope_deltas=f.rope_deltas, logits_to_keep=1)
    rows = [out.logits[0, -1].clone()]
    for y in forced:
        rows.append(lm(np.array([[int(y)]]), cache=cache).logits[0, -1].clone())
    n = cache[0].offset
    cache[0]._seq.release()
    return torch.stack(rows), f, n


def _check_rows(got, ref, tol_rms, tag):
    """every row: rel-rms error below tol_rms; argmax identical wherever the oracle's top-2 margin exceeds 4 x the
    per-element error bound (0.25 rms is far above it for every tolerance used here)"""
    worst = 0.0
    for i in range(ref.shape[0]):
        e = _rel_rms(got[i], ref[i])
        worst = max(worst, e)
        assert e < tol_rms, (tag, i, e)
        r = ref[i].float()
        top2 = r.topk(2).values
        if float(top2[0] - top2[1]) > 0.25 * float(r.pow(2).mean().sqrt()):
            assert int(got[i].float().argmax()) == int(r.argmax()), (tag, i)
    return worst


# ------------------------------------------------------------------------------------------------ tiny, every step
@pytest.fixture(scope="module")
def tiny():
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    return cfg, W, build_product_model(cfg, W, kv_pool_tokens=8192, max_seqs=16)


@pytest.mark.parametrize("sizes", [[(56, 84)], [(56, 56), (84, 56)], []])
def test_teacher_forced_decode_logits_every_step(tiny, sizes):
    cfg, W, model = tiny
    ids, pix, thw = synth_request(cfg, sizes, n_text=14, seed=40 + len(sizes)) if sizes else \
        (np.random.default_rng(41).integers(3, 1000, (1, 23)), None, None)
    forced = np.random.default_rng(42).integers(3, 1000, 70)       # crosses the 64-token KV page boundary
    ref = oq.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pix).to(BF) if pix is not None else None, thw, forced)
    got, f, n = _engine_teacher_forced(model, ids, pix, thw, forced)
    assert n == ids.shape[1] + len(forced)
    assert got.shape == ref.shape == (1 + len(forced), cfg.text.vocab_size)
    # 2 layers of bf16: 2e-2 of the logit rms per row; every row, not only while tokens agree
    worst = _check_rows(got, ref, 2e-2, "tiny")
    # every logit of every step as a bf16 value, element-wise: 4 ulps + 8 % of the rms.  Measured on MI355X: 13 of
    # 72,704 elements beyond 4 ulps + 3 %, the worst 0.19 at rms 3.2 (5.9 %) - single bf16 flips of an activation that
    # the next layer amplifies; a wrong position, page or head shows up as O(1) x rms
    ok, rep = bf16_close(got, ref, ulps=4, atol_rms=8e-2)
    assert ok, rep
    print(f"tiny teacher-forced {len(sizes)} image(s): worst row rel-rms {worst:.4f}; {rep}")


def _peaked_tiny():
    cfg = oq.tiny_cfg()
    cfg.text.tie_word_embeddings = False
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    for k in list(W):      # halve the residual branches so the input embedding stays visible to the head
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            W[k] = (W[k].float() * 0.5).to(BF)
    return cfg, oq.peak_head(W, cfg, gamma=4.0, stride=389, n_cycle=1000)


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("sizes", [[(56, 84)], []])
def test_peaked_head_greedy_tokens_identical_no_tie_rule(sizes, use_graph):
    from mlx_vlm_amd.generate import generate_step

    cfg, W = _peaked_tiny()
    model = build_product_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    ids, pix, thw = synth_request(cfg, sizes, n_text=14, seed=50) if sizes else \
        (np.random.default_rng(51).integers(3, 1000, (1, 17)), None, None)
    n_new = 40
    ref_toks, ref_logits = oq.generate_greedy(W, cfg, ids, torch.from_numpy(pix).to(BF) if pix is not None else None, thw,
                                              max_tokens=n_new, return_logits=True)
    # the construction must hold in the ORACLE before anything is asked of the engine: a walk of the 1000-cycle with
    # every top-2 margin above half the logit rms (no tie band anywhere near)
    last = int(ids[0, -1])
    assert ref_toks == [(last + 389 * (i + 1)) % 1000 for i in range(n_new)]
    r = ref_logits.float()
    top2 = r.topk(2, dim=-1).values
    assert float(((top2[:, 0] - top2[:, 1]) / r.pow(2).mean(-1).sqrt()).min()) > 0.5
    kw = dict(image_grid_thw=thw) if thw is not None else {}
    toks, lps = [], []
    for t, lp in generate_step(ids, model, torch.from_numpy(pix) if pix is not None else None, None, max_tokens=n_new,
                               temperature=0.0, use_graph=use_graph, lookahead=4, **kw):
        toks.append(t)
        lps.append(lp.float().cpu())
    assert toks == ref_toks                                        # identical, no escape
    ref_lp = O.logprobs_from_logits(ref_logits)
    for i in range(n_new):                                         # every step's bf16 log-probs: 2 ulps + 3 % rms
        ok, rep = bf16_close(lps[i], ref_lp[i], ulps=2, atol_rms=3e-2)
        assert ok, (i, rep)


# ------------------------------------------------------------------------------------------------ BASELINE configs[1]
def test_full_depth_qwen2_vl_2b_image_prefill_and_teacher_forced_decode():
    """Qwen2-VL-2B at FULL size (the benchmarked model: 32 ViT blocks of 1280, 28 decoder layers of 1536 / 8960, GQA
    12:2, V = 151,936, tied head), one 448 x 448 image (1024 patches -> 256 image tokens) + 128 text tokens.
    Random-init weights (std 0.02, as bench.py); the oracle runs the same bf16 typed graph on the host (~1 min)."""
    cfg = oq.Cfg()                                                  # defaults = 2B dims
    W = oq.random_weights(cfg, seed=7, dtype=BF, std=0.02)
    model = build_product_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (3, 448, 448), dtype=np.uint8)
    from oracle import image_processor as oip

    pix, thw = oip.process([img])
    n_img = int(thw.prod()) // 4
    assert (pix.shape[0], n_img) == (1024, 256)
    text = np.random.default_rng(1003).integers(0, 151643, 128)
    ids = np.concatenate([[cfg.vision_start_token_id], np.full(n_img, cfg.image_token_id),
                          [cfg.vision_start_token_id + 1], text]).astype(np.int64)[None]
    forced = np.random.default_rng(1004).integers(0, 151643, 8)
    pix_t = torch.from_numpy(pix).to(BF)
    ref_feats = oq.vision_tower(W, cfg, pix_t, thw)
    ref = oq.decode_teacher_forced(W, cfg, ids, pix_t, thw, forced)
    feats = model.vision_tower(torch.from_numpy(pix), thw)
    # 32 blocks + merger in bf16: 3e-2 of the feature rms
    e_feat = _rel_rms(feats, ref_feats)
    assert e_feat < 3e-2, e_feat
    got, f, n = _engine_teacher_forced(model, ids, pix, thw, forced)
    assert n == ids.shape[1] + len(forced) == 386 + 8
    # rope index of the image prompt: integer-exact
    opos, odelta = oq.get_rope_index(cfg, ids, thw)
    np.testing.assert_array_equal(np.asarray(f.position_ids), np.asarray(opos))
    np.testing.assert_array_equal(np.asarray(f.rope_deltas), np.asarray(odelta))
    # 28 layers of bf16 after a 32-block tower: 4e-2 of the logit rms on every row (prefill row + 8 decode steps)
    worst = _check_rows(got, ref, 4e-2, "2B")
    print(f"full-depth 2B: feature rel-rms {e_feat:.4f}, worst logit-row rel-rms {worst:.4f}")


# ------------------------------------------------------------------------------------------------ BASELINE configs[0]
def test_full_size_nanollava_image_prefill_and_teacher_forced_decode():
    """nanoLLaVA at FULL size: SigLIP-so400m/14-384 tower (27 layers of 1152 / 4304, 16 heads of 72, 729 patches),
    mlp2x_gelu projector, Qwen1.5-0.5B (24 layers of 1024 / 2816, 16 heads of 64, V = 151,936, tied)."""
    from oracle import llava_bunny as ob
    from tests.helpers import build_bunny_model

    cfg = ob.Cfg(text=ob.TextCfg(), vision=ob.VisionCfg())          # defaults = real dims
    assert (cfg.text.hidden_size, cfg.text.num_hidden_layers, cfg.vision.hidden_size, cfg.vision.num_hidden_layers) == \
        (1024, 24, 1152, 27)
    W = ob.random_weights(cfg, seed=11, dtype=BF, std=0.02, embed_std=0.02)
    model = build_bunny_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (336, 336, 3), dtype=np.uint8)       # BASELINE: 336 x 336 in, resized to 384 x 384
    pix = torch.from_numpy(ob.preprocess([img]))
    text = rng.integers(0, 151643, 64)
    ids = np.concatenate([text[:20], [cfg.image_token_index], text[20:]]).astype(np.int64)[None]
    forced = rng.integers(0, 151643, 8)
    ref_last = ob.vision_tower(W, cfg, pix.to(BF))
    last = model.vision_tower(pix)
    e_tower = _rel_rms(last, ref_last[0])
    assert e_tower < 3e-2, e_tower                                  # 27 layers of bf16
    ref = ob.decode_teacher_forced(W, cfg, ids, pix, forced)
    lm = model.language_model
    f = model.get_input_embeddings(ids, pix)
    L = f.inputs_embeds.shape[1]
    assert L == ids.shape[1] + 728
    from mlx_vlm_amd.models import cache as cache_mod

    cache = cache_mod.make_prompt_cache(lm)
    rows = [lm.prefill(f.inputs_embeds.reshape(L, -1), np.asarray(f.position_ids).reshape(3, L), [cache], [L], "last")[0].clone()]
    for y in forced:
        rows.append(lm(np.array([[int(y)]]), cache=cache).logits[0, -1].clone())
    cache[0]._seq.release()
    got = torch.stack(rows)
    worst = _check_rows(got, ref, 3e-2, "nanoLLaVA")                # 3e-2 rel-rms, as everywhere else
    print(f"full-size nanoLLaVA: tower rel-rms {e_tower:.4f}, worst logit-row rel-rms {worst:.4f}")
    # ... and the distance to what the reference REALLY outputs for BASELINE configs[0]: llava_bunny never casts the pixels to the
    # weight dtype (reference mlx_vlm/models/llava_bunny/llava_bunny.py:57-120), so with a bf16 checkpoint MLX's type promotion
    # carries FLOAT32 activations from the patch embedding through the tower, the projector, the prompt pass and the float32
    # cache (oracle: cast_pixels=False, pinned against the reference's own files in tests/test_oracle_ref_golden_bunny.py).  The
    # engine computes the bf16 typed graph (deviation (i)); this is that deviation as a number, next to the
    # distance of the ORACLE's bf16 graph from the same fp32-activation run (the part of it that is bf16 activations as such).
    ref32 = ob.decode_teacher_forced(W, cfg, ids, pix, forced, cast_pixels=False)
    d_engine = max(_rel_rms(got[i], ref32[i]) for i in range(len(ref32)))
    d_typed = max(_rel_rms(ref[i], ref32[i]) for i in range(len(ref32)))
    same = sum(int(got[i].float().argmax()) == int(ref32[i].float().argmax()) for i in range(len(ref32)))
    print(f"full-size nanoLLaVA vs the reference's fp32-activation run: engine {d_engine:.4f}, the oracle's bf16 graph {d_typed:.4f} "
          f"(worst logit-row rel-rms over prefill + {len(forced)} forced steps); greedy token equal on {same} / {len(ref32)} rows")
    assert d_engine < 5e-2, d_engine                                # bf16 activations through 27 + 24 layers vs float32 ones
    assert d_engine < 1.5 * d_typed + 1e-2, (d_engine, d_typed)     # ... and no more than the typed graph itself is away from it
