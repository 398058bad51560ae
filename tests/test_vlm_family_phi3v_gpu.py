"""Phi-3.5-vision (`phi3_v`, SURVEY §8f row 2 / BASELINE configs[4]) on MI355X: the HIP path against the oracle
(oracle/phi3_v.py, pinned to the reference's own files by tests/test_oracle_ref_golden_phi3v.py).

  * CLIP tower + HD transform + projection: feature rows of every image vs the oracle (views of different counts in one
    batch, separators, the reference's plain-reshape arrangement of the local views);
  * prefill + teacher-forced decode (a seeded random token stream, every step's logits), bf16 and MLX 4-bit weights;
    Su-scaled RoPE with the typed q / k scale in the prefill rope pass, the 1-row v_dot2c qkv epilogue, the 4-bit one and
    the 16-row MFMA one;
  * greedy generate_step through the captured graph, 16 decode rows through the BatchGenerator, load() from an HF-layout
    checkpoint on disk;
  * real Phi-3.5-vision widths (3072 / 32 heads of 96 / 8192 / V 32064, CLIP ViT-L) at reduced depth, 4-bit.
"""
import json

import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import phi3_v as op
from oracle import quant as Q
from tests.helpers import bf16_close, build_phi3v_model

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
# weight scales of the tiny model in these tests: at std 0.1 (the golden generator's, chosen for varied greedy tokens) the
# ORACLE's own bf16-vs-f32 distance is 3-5 % of the logit rms - attention scores are large and every bf16 flip of q / k is
# amplified - so a 2e-2 bound would measure noise; at 0.04 that distance is 1 % and dropping e.g. the Su scale still moves
# the logits by 15-30 % (both measured on the host)
SCALES = dict(std=0.04, embed_std=0.2)


def _rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def _request(cfg, images, n_text=(5, 4, 6), seed=0, vocab_hi=1000):
    """-> (input_ids [1, L] with runs of -1, -2, ..., pixel_values f32 [B, T, 3, 336, 336], image_sizes [B, 2])"""
    rng = np.random.default_rng(seed)
    pv, sz = op.preprocess(images) if images else (None, None)
    parts = [rng.integers(3, vocab_hi, n_text[0])]
    for j, im in enumerate(images):
        parts += [np.full(op.num_image_tokens(im.shape[1], im.shape[0]), -(j + 1)), rng.integers(3, vocab_hi, n_text[1 + j % 2])]
    return np.concatenate(parts).astype(np.int64)[None], pv, sz


def _images(seed, shapes):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]


def _engine_teacher_forced(model, ids, pv, sz, forced):
    lm = model.language_model
    kw = dict(image_sizes=sz) if pv is not None else {}
    f = model.get_input_embeddings(ids, torch.from_numpy(pv) if pv is not None else None, **kw)
    cache = lm.make_cache()
    out = lm(ids, f.inputs_embeds, cache=cache, position_ids=f.position_ids, rope_deltas=f.rope_deltas, logits_to_keep=1)
    rows = [out.logits[0, -1].clone()]
    for y in forced:
        rows.append(lm(np.array([[int(y)]]), cache=cache).logits[0, -1].clone())
    n = cache[0].offset
    cache[0]._seq.release()
    return torch.stack(rows), f, n


def _check_rows(got, ref, tol_rms, tag):
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


@pytest.fixture(scope="module")
def tiny():
    cfg = op.tiny_cfg()
    W = op.random_weights(cfg, seed=4321, dtype=BF, **SCALES)
    return cfg, W, build_phi3v_model(cfg, W, kv_pool_tokens=16384, max_seqs=40)


def test_clip_tower_hd_transform_and_projection_vs_oracle(tiny):
    """Two images with different view counts in one call (5 and 4 views, the second zero-padded by the processor): every
    projected row - local rows, sub_GN separators, glb_GN, global rows - against the oracle: 2e-2 rel-rms per image
    (3 CLIP layers + 2 projection GEMMs in bf16), separator rows (which never see the tower) to 2 ulps."""
    cfg, W, model = tiny
    imgs = _images(3, [(120, 200), (300, 90)])
    pv, sz = op.preprocess(imgs)
    assert pv.shape[:2] == (2, 5) and sz.tolist() == [[672, 672], [1008, 336]]
    ref = op.image_features(W, cfg, torch.from_numpy(pv).to(BF), sz)
    got = model.vision_model.image_features(torch.from_numpy(pv), sz)
    assert [tuple(g.shape) for g in got] == [tuple(r.shape) for r in ref]
    for b, (g, r) in enumerate(zip(got, ref)):
        assert g.shape[0] == op.num_image_tokens(imgs[b].shape[1], imgs[b].shape[0])
        assert _rel_rms(g, r) < 2e-2, (b, _rel_rms(g, r))
    # the glb_GN row: a projection of a parameter alone (no tower): 2 ulps
    h, w = 2, 2
    at = h * 12 * (w * 12 + 1)
    ok, rep = bf16_close(got[0][at], ref[0][at], ulps=2, atol_rms=5e-3)
    assert ok, rep
    # CLIP features alone (encoder_states[-2] without the class row), all views of image 0
    feat = model.vision_model.clip_features(torch.from_numpy(pv[0]))
    ref_feat = op.clip_features(W, cfg, torch.from_numpy(pv[0]).to(BF))
    assert feat.shape == ref_feat.shape == (5, 576, 1024) and _rel_rms(feat, ref_feat) < 1.5e-2


@pytest.mark.parametrize("n_images", [1, 2, 0])
def test_teacher_forced_decode_logits_every_step(tiny, n_images):
    """prefill (Su-scaled RoPE through vlm_mrope_kvwrite_scaled) + 70 forced decode steps (the fused qkv epilogue with the
    typed q / k scale; crosses the 64-token KV page boundary): every row 2e-2 rel-rms, argmax wherever the oracle's margin
    allows, every element 4 ulps + 8 % rms."""
    cfg, W, model = tiny
    imgs = _images(10 + n_images, [(336, 336), (100, 260)][:n_images])
    ids, pv, sz = _request(cfg, imgs, seed=20 + n_images)
    forced = np.random.default_rng(42).integers(3, 1000, 70)
    ref = op.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pv) if pv is not None else None, sz, forced)
    got, f, n = _engine_teacher_forced(model, ids, pv, sz, forced)
    assert n == ids.shape[1] + len(forced) and got.shape == ref.shape
    worst = _check_rows(got, ref, 2e-2, "phi3v tiny")
    ok, rep = bf16_close(got, ref, ulps=4, atol_rms=8e-2)
    assert ok, rep
    print(f"phi3v tiny teacher-forced, {n_images} image(s): worst row rel-rms {worst:.4f}; {rep}")


def test_su_rope_scale_is_applied_with_the_reference_rounding(tiny):
    """The q / k scale is not cosmetic: the same engine with rope_qk_scale left at 1 (scale dropped) must DISAGREE with
    the oracle by far more than the tolerance used above - i.e. the tests above can see the scale."""
    from tests.helpers import phi3v_config_from_oracle
    from mlx_vlm_amd.models.phi3_v import Model

    cfg, W, _ = tiny
    mc = phi3v_config_from_oracle(cfg)
    mc.max_position_embeddings = mc.original_max_position_embeddings            # factor 1: SuScaledRoPE's scale becomes 1
    model = Model(mc, kv_pool_tokens=4096, max_seqs=4)
    model.load_weights(W)
    assert model.language_model.args.rope_qk_scale is None
    ids, _, _ = _request(cfg, [], seed=31)
    forced = np.random.default_rng(43).integers(3, 1000, 6)
    ref = op.decode_teacher_forced(W, cfg, ids, None, None, forced)
    got, _, _ = _engine_teacher_forced(model, ids, None, None, forced)
    assert min(_rel_rms(got[i], ref[i]) for i in range(ref.shape[0])) > 0.1          # (host measurement: 15-30 %)


def test_generate_step_greedy_graph_and_eager_match_oracle_until_a_tie(tiny):
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    imgs = _images(50, [(336, 336)])
    ids, pv, sz = _request(cfg, imgs, seed=51)
    n_new = 12
    ref_toks, ref_logits = op.generate_greedy(W, cfg, ids, torch.from_numpy(pv), sz, max_tokens=n_new, return_logits=True)
    ref_lp = O.logprobs_from_logits(ref_logits)
    for use_graph in (True, False):
        toks, lps = [], []
        for t, lp in generate_step(ids, model, torch.from_numpy(pv), None, max_tokens=n_new, temperature=0.0, use_graph=use_graph,
                                   image_sizes=sz):
            toks.append(t)
            lps.append(lp.float().cpu())
        assert len(toks) == n_new
        for i in range(n_new):
            if toks[i] != ref_toks[i]:      # a tie inside bf16 noise: the oracle's own top-2 margin must be tiny
                r = ref_logits[i].float()
                top2 = r.topk(2).values
                assert float(top2[0] - top2[1]) < 0.06 * float(r.pow(2).mean().sqrt()), (use_graph, i, toks, ref_toks)
                break
            ok, rep = bf16_close(lps[i], ref_lp[i], ulps=2, atol_rms=3e-2)
            assert ok, (use_graph, i, rep)
        assert toks[0] == ref_toks[0]


def _quantized_tiny(seed=4321):
    cfg = op.tiny_cfg()
    W = op.random_weights(cfg, seed=seed, dtype=BF, **SCALES)
    ck, ow = Q.quantize_checkpoint(W, predicate=lambda p, v: not p.startswith("model.vision_embed_tokens."))
    return cfg, ck, ow


def test_quantized_4bit_teacher_forced_decode_vs_oracle():
    """MLX 4-bit language model (embedding, fused qkv / gate_up rows split and re-laid, o_proj columns moved group by
    group, head): prefill through dequant + bf16 GEMM, decode through the fused 4-bit GEMVs, against the oracle running
    the SAME packed weights through nn.QuantizedLinear restated."""
    cfg, ck, ow = _quantized_tiny()
    model = build_phi3v_model(cfg, ck, kv_pool_tokens=4096, max_seqs=4)
    assert model.language_model.quantized
    imgs = _images(60, [(200, 336)])
    ids, pv, sz = _request(cfg, imgs, seed=61)
    forced = np.random.default_rng(62).integers(3, 1000, 70)
    ref = op.decode_teacher_forced(ow, cfg, ids, torch.from_numpy(pv), sz, forced)
    got, _, n = _engine_teacher_forced(model, ids, pv, sz, forced)
    assert n == ids.shape[1] + len(forced)
    worst = _check_rows(got, ref, 2e-2, "phi3v 4-bit")
    print(f"phi3v 4-bit teacher-forced: worst row rel-rms {worst:.4f}")


@pytest.mark.parametrize("w4", [False, True])
def test_16_row_batch_equals_single_requests(w4):
    """BASELINE configs[4] runs batch 16: 18 requests through 16 decode rows (qkv + Su-RoPE + KV write and every projection
    on the matrix cores, csrc/gemv_mfma.hip) vs the same requests alone (v_dot2c GEMVs): tokens equal except at ties, token
    log-probs within 2 ulps of the step's log-prob span."""
    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.generate import generate_step

    if w4:
        cfg, ck, _ = _quantized_tiny()
    else:
        cfg = op.tiny_cfg()
        ck = op.random_weights(cfg, seed=4321, dtype=BF, **SCALES)
    model = build_phi3v_model(cfg, ck, kv_pool_tokens=32768, max_seqs=40)
    reqs = []
    for i in range(18):
        imgs = _images(100 + i, [(336, 336)]) if i % 3 == 0 else []
        reqs.append(_request(cfg, imgs, n_text=(4 + i % 5, 3, 5), seed=200 + i))
    max_tokens = [5 + (3 * i) % 7 for i in range(18)]
    singles = []
    for (ids, pv, sz), m in zip(reqs, max_tokens):
        kw = dict(image_sizes=sz) if pv is not None else {}
        singles.append([(t, float(lp[t]), float(lp.float().abs().max())) for t, lp in
                        generate_step(ids, model, torch.from_numpy(pv) if pv is not None else None, None, max_tokens=m, **kw)])
    gen = BatchGenerator(model, None, max_tokens=8, completion_batch_size=16, prefill_batch_size=8)
    assert gen.completion_batch_size == 16
    kws = [dict(pixel_values=torch.from_numpy(pv), image_sizes=sz) if pv is not None else {} for _, pv, sz in reqs]
    uids = gen.insert([r[0].reshape(-1) for r in reqs], list(max_tokens), prompt_kwargs=kws)
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
    assert n_equal >= 0.8 * sum(max_tokens), (n_equal, sum(max_tokens))


def test_load_from_hf_layout_checkpoint_and_generate(tmp_path):
    """config.json + safetensors under the HF names (conv weight in torch layout, a position_ids buffer) + a tokenizer ->
    load() -> prepare_inputs -> generate_step: same tokens as the model built in memory."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from mlx_vlm_amd import utils
    from mlx_vlm_amd.generate import generate_step
    from tests.helpers import phi3v_config_from_oracle

    cfg = op.tiny_cfg()
    W = op.random_weights(cfg, seed=9, dtype=BF, **op.TEST_WEIGHT_SCALES)
    hf = {k: v.contiguous() for k, v in W.items()}
    pk = op.CLIP + "embeddings.patch_embedding.weight"
    hf[pk] = W[pk].permute(0, 3, 1, 2).contiguous()                      # torch conv layout (O, C, kH, kW)
    hf[op.CLIP + "embeddings.position_ids"] = torch.arange(577)[None]
    save_file(hf, str(tmp_path / "model.safetensors"))
    conf = {k: v for k, v in phi3v_config_from_oracle(cfg).to_dict().items() if k not in ("text_config", "vision_config")}
    conf["vision_config"] = dict(num_hidden_layers=cfg.vision.num_hidden_layers, intermediate_size=cfg.vision.intermediate_size)
    (tmp_path / "config.json").write_text(json.dumps(conf))
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2, **{f"w{i}": i + 3 for i in range(900)}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>").save_pretrained(str(tmp_path))
    model, proc = utils.load(str(tmp_path), kv_pool_tokens=4096, max_seqs=4)
    assert type(model).__module__.endswith("phi3_v.phi3_v")
    im = _images(70, [(336, 336)])[0]
    inp = utils.prepare_inputs(proc, images=[im], prompts="w5 w9 <|image_1|> w7 w30 w2 w11")
    mem = build_phi3v_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    a = [t for t, _ in generate_step(inp["input_ids"], model, torch.from_numpy(inp["pixel_values"]), None, max_tokens=6,
                                     image_sizes=inp["image_sizes"])]
    b = [t for t, _ in generate_step(inp["input_ids"], mem, torch.from_numpy(inp["pixel_values"]), None, max_tokens=6,
                                     image_sizes=inp["image_sizes"])]
    assert a == b and len(a) == 6


def test_real_widths_4bit_two_layers_vs_oracle():
    """Phi-3.5-vision's real widths (hidden 3072, 32 heads of 96 -> o_proj K = 4096 in the engine layout, intermediate 8192,
    V = 32064, CLIP ViT-L 1024 / 4096) at 2 decoder layers and 3 CLIP layers, MLX 4-bit language model, a 336 x 336 image
    (757 image tokens at num_crops 4): image features, last-row prefill logits and 6 teacher-forced decode steps."""
    short, long = op.su_factors(96, seed=9)
    cfg = op.Cfg(text=op.TextCfg(num_hidden_layers=2, short_factor=short, long_factor=long), vision=op.VisionCfg(num_hidden_layers=3))
    W = op.random_weights(cfg, seed=13, dtype=BF, std=0.02, embed_std=0.02)
    ck, ow = Q.quantize_checkpoint(W, predicate=lambda p, v: not p.startswith("model.vision_embed_tokens."))
    model = build_phi3v_model(cfg, ck, kv_pool_tokens=4096, max_seqs=4)
    imgs = _images(80, [(336, 336)])
    ids, pv, sz = _request(cfg, imgs, n_text=(20, 40, 40), seed=81, vocab_hi=32000)
    assert int((ids < 0).sum()) == 757
    forced = np.random.default_rng(82).integers(3, 32000, 6)
    ref_rows = op.image_features(ow, cfg, torch.from_numpy(pv).to(BF), sz)[0]
    got_rows = model.vision_model.image_features(torch.from_numpy(pv), sz)[0]
    assert _rel_rms(got_rows, ref_rows) < 2e-2
    ref = op.decode_teacher_forced(ow, cfg, ids, torch.from_numpy(pv), sz, forced)
    got, _, n = _engine_teacher_forced(model, ids, pv, sz, forced)
    assert n == ids.shape[1] + 6
    worst = _check_rows(got, ref, 2e-2, "phi3.5 widths 4-bit")
    print(f"phi3.5 real widths, 4-bit: worst row rel-rms {worst:.4f}")


@pytest.mark.parametrize("prompt_len", [4090, 4200])
def test_su_rope_long_factor_regime_switches_per_call(tiny, prompt_len):
    """SuScaledRoPE picks its factors per call (rope_utils.py:168-172): long iff cache offset + tokens of the call exceed
    original_max_position_embeddings (4096).  4090-token prompt: short-factor prefill, then teacher-forced decode steps whose
    cache offset runs 4090 .. 4101 - the steps from offset 4096 on use the long factors over keys cached with the short
    ones; 4200-token prompt: long from the prefill on.  Every row vs the oracle (which restates that rule and is pinned to the
    reference's rope on both sides of the limit); through the module contract and through generate_step's graph replay."""
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    ids = np.random.default_rng(300 + prompt_len).integers(3, 1000, (1, prompt_len))
    forced = np.random.default_rng(301).integers(3, 1000, 12)
    ref = op.decode_teacher_forced(W, cfg, ids, None, None, forced)
    got, _, n = _engine_teacher_forced(model, ids, None, None, forced)
    assert n == prompt_len + 12
    worst = _check_rows(got, ref, 2.5e-2, f"phi3v long regime {prompt_len}")
    # the same through generate_step (lookahead batches of graph replays split at the crossing): greedy tokens vs the oracle
    ref_toks, ref_logits = op.generate_greedy(W, cfg, ids, None, None, max_tokens=12, return_logits=True)
    toks = [t for t, _ in generate_step(ids, model, None, None, max_tokens=12, temperature=0.0, lookahead=4)]
    for i in range(12):
        if toks[i] != ref_toks[i]:
            r = ref_logits[i].float()
            top2 = r.topk(2).values
            assert float(top2[0] - top2[1]) < 0.06 * float(r.pow(2).mean().sqrt()), (i, toks, ref_toks)
            break
    assert toks[0] == ref_toks[0]
    # back to a short prompt afterwards: the table must have been switched back
    ids2, _, _ = _request(cfg, [], seed=77)
    f2 = np.random.default_rng(78).integers(3, 1000, 4)
    _check_rows(_engine_teacher_forced(model, ids2, None, None, f2)[0], op.decode_teacher_forced(W, cfg, ids2, None, None, f2), 2e-2,
                "phi3v short again")
    print(f"phi3v long-factor regime, prompt {prompt_len}: worst row rel-rms {worst:.4f}")


@pytest.mark.parametrize("B", [2, 9, 20])
@pytest.mark.parametrize("w4", [False, True])
def test_su_rope_regime_of_a_batched_decode_step_follows_its_longest_row(w4, B):
    """SuScaledRoPE decides per CALL (rope_utils.py:168-172): position_end = max(cache offset over the rows) + 1.  Two rows
    decode together: row 0 stays near offset 30, row 1 runs from offset 4090 across 4096 - from the step where row 1's
    offset is 4096 on, BOTH rows rotate with the long factors (row 0's new keys too), exactly as the reference's batched
    call does (pinned in test_oracle_ref_golden_phi3v.py against the reference's own class run with an offset array).
    B = 2: the v_dot2c qkv epilogues (bf16 / 4-bit); B = 9: the skinny-M MFMA form (8 short rows + the long one); B = 20: a
    WIDE step (prefill GEMMs + vlm_mrope_kvwrite_decode, which decides the regime from the rows' slots on the device).
    The engine evaluates the rule inside the qkv epilogue from the rows' cache offsets (vlm_llm_config.rope_long_from):
    every row of every step against the oracle with the call-wide position_end; and a control - row 0 decoded ALONE over
    the same steps stays short and must differ from its batched logits after the crossing."""
    if w4:
        cfg, ck, ow = _quantized_tiny(seed=777)
    else:
        cfg = op.tiny_cfg()
        ck = ow = op.random_weights(cfg, seed=777, dtype=BF, **SCALES)
    model = build_phi3v_model(cfg, ck, kv_pool_tokens=16384, max_seqs=max(16, B + 4))
    lm = model.language_model
    lim = cfg.text.original_max_position_embeddings
    rng = np.random.default_rng(910)
    prompts = [rng.integers(3, 1000, 30 + 7 * r).astype(np.int64) for r in range(B - 1)] + [rng.integers(3, 1000, lim - 6).astype(np.int64)]
    n_steps = 10
    forced = rng.integers(3, 1000, (n_steps, B))
    caches, ocaches = [], []
    for p in prompts:
        c = lm.make_cache()
        lm(p[None], cache=c, logits_to_keep=1)
        caches.append(c)
        oc = [O.KVCache() for _ in range(cfg.text.num_hidden_layers)]
        op.language_model(ow, cfg, op.embed_tokens(ow, p[None]), oc, last_only=True)
        ocaches.append(oc)
    worst, crossed = 0.0, 0
    row0_batched = []
    for s in range(n_steps):
        pe = max(len(p) + s for p in prompts) + 1                      # the call's position_end
        crossed += pe > lim
        got = lm(forced[s].reshape(B, 1), cache=caches).logits[:, 0]
        row0_batched.append(got[0].float().cpu())
        for r in range(B):
            ref = op.language_model(ow, cfg, op.embed_tokens(ow, np.array([[int(forced[s, r])]])), ocaches[r], position_end=pe)[0, 0]
            e = _rel_rms(got[r], ref)
            worst = max(worst, e)
            assert e < 2.5e-2, (s, r, pe, e)
    assert 0 < crossed < n_steps                                        # steps on both sides of the limit
    for c in caches:
        c[0]._seq.release()
    # control: row 0 alone never leaves the short regime - same tokens, different logits once the batch has crossed
    c0 = lm.make_cache()
    lm(prompts[0][None], cache=c0, logits_to_keep=1)
    alone = [lm(np.array([[int(forced[s, 0])]]), cache=c0).logits[0, 0].float().cpu() for s in range(n_steps)]
    c0[0]._seq.release()
    first_long = n_steps - crossed
    assert _rel_rms(alone[first_long - 1], row0_batched[first_long - 1]) < 2.5e-2      # before: the same computation
    assert _rel_rms(alone[n_steps - 1], row0_batched[n_steps - 1]) > 5e-2               # after: another rope regime
    print(f"batched Su-RoPE regime (w4={w4}, B={B}): worst row rel-rms {worst:.4f}, {crossed} long steps of {n_steps}")


def test_batch_generator_row_crossing_the_su_rope_limit():
    """A continuous batch whose row crosses original_max_position_embeddings used to raise; it now decodes, the regime of
    every step being decided on the device by the longest live row.  One request alone in the generator (the batch IS the
    sequence): tokens equal generate_step's."""
    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.generate import generate_step

    cfg = op.tiny_cfg()
    W = op.random_weights(cfg, seed=778, dtype=BF, **SCALES)
    model = build_phi3v_model(cfg, W, kv_pool_tokens=16384, max_seqs=40)
    lim = cfg.text.original_max_position_embeddings
    ids = np.random.default_rng(911).integers(3, 1000, (1, lim - 5))
    single = [t for t, _ in generate_step(ids, model, None, None, max_tokens=12, temperature=0.0)]
    gen = BatchGenerator(model, None, max_tokens=12, completion_batch_size=4)
    (uid,) = gen.insert([ids.reshape(-1)], [12])
    got = []
    while gen.has_work:
        _, out = gen.next()
        got += [r.token for r in out if r.uid == uid]
    gen.close()
    assert got == single and len(got) == 12
