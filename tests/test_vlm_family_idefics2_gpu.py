"""Idefics2 (SURVEY §8f row 3 / BASELINE configs[3]) on MI355X: the HIP path against the oracle (oracle/idefics2.py, pinned
to the reference's own files by tests/test_oracle_ref_golden_idefics2.py).

  * SigLIP tower (bucketed position ids, images padded to a common size, padding images dropped) + modality projection +
    perceiver resampler (GQA cross-attention of the latents over [context | latents] on the varlen flash-attention kernel);
  * multi-image prompts: masked_scatter, prefill + teacher-forced decode, every step's logits;
  * greedy generate_step (graph and eager), continuous batching with per-request images, load() from an HF-layout checkpoint;
  * Mistral-7B / Idefics2-8B widths: hidden 4096 through the K = 4096 norm-prologue GEMVs, vocabulary 32003 (not a multiple
    of 8: padded logits pitch), SigLIP-so400m 1152 / 4304 with 72-wide heads, a 4 x 336 x 336 prompt (BASELINE configs[3]).
"""
import json

import numpy as np
import pytest
import torch

from oracle import idefics2 as oi
from oracle import ops as O
from tests.helpers import bf16_close, build_idefics2_model

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
SMALL = dict(shortest_edge=56, longest_edge=140)            # the tiny tower has a 10 x 10 position table (140 px)


def _rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def _images(seed, shapes):
    rng = np.random.default_rng(seed)
    return [rng.integers(1, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]


def _request(cfg, images, seed=0, n_text=(5, 3, 4), vocab_hi=1000, **proc_kw):
    """one prompt with len(images) images -> (input_ids [1, L], pixel_values [1, N, 3, H, W] f32, pixel_attention_mask)"""
    rng = np.random.default_rng(seed)
    nl = cfg.perceiver.resampler_n_latents
    parts = [rng.integers(3, vocab_hi, n_text[0])]
    for j in range(len(images)):
        parts += [np.full(nl, cfg.image_token_id), rng.integers(3, vocab_hi, n_text[1 + j % 2])]
    ids = np.concatenate(parts).astype(np.int64)[None]
    if not images:
        return ids, None, None
    pv, pm = oi.preprocess([images], **proc_kw)
    return ids, pv, pm


def _engine_teacher_forced(model, ids, pv, pm, forced):
    lm = model.language_model
    kw = dict(pixel_attention_mask=pm) if pv is not None else {}
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
    cfg = oi.tiny_cfg()
    W = oi.random_weights(cfg, seed=4321, dtype=BF, **oi.TEST_WEIGHT_SCALES)
    return cfg, W, build_idefics2_model(cfg, W, kv_pool_tokens=16384, max_seqs=40)


def test_tower_and_connector_vs_oracle(tiny):
    """Three real images of two samples (different sizes -> padding and patch masks; one all-zero padding image dropped): the
    tower's pooler output and the resampler outputs against the oracle."""
    cfg, W, model = tiny
    a, b, c = _images(3, [(90, 60), (56, 70), (70, 70)])
    pv, pm = oi.preprocess([[a, b], [c]], **SMALL)
    assert pv.shape[:2] == (2, 2)
    real, pmask = oi.real_images_and_patch_mask(torch.from_numpy(pv), pm, cfg.vision.patch_size)
    assert real.shape[0] == 3
    ref_pooled = oi.vision_tower(W, cfg, real.to(BF), pmask)
    own_real, own_mask = model._real_images(pv, pm)
    pooled = model.vision_model(torch.from_numpy(own_real), own_mask)
    assert _rel_rms(pooled, ref_pooled.reshape(-1, ref_pooled.shape[-1])) < 1.5e-2
    ref = oi.image_features(W, cfg, torch.from_numpy(pv).to(BF), pm)
    got = model.encode_image(torch.from_numpy(pv), pm)
    assert got.shape == (3 * cfg.perceiver.resampler_n_latents, cfg.text.hidden_size)
    e = _rel_rms(got, ref.reshape(-1, ref.shape[-1]))
    assert e < 2e-2, e
    # the connector alone from the oracle's pooled states: isolates the perceiver's cross-attention layout
    got_c = model.connector(ref_pooled.reshape(-1, ref_pooled.shape[-1]).cuda().contiguous(), 3)
    ref_c = oi.connector(W, cfg, ref_pooled)
    assert _rel_rms(got_c, ref_c.reshape(-1, ref_c.shape[-1])) < 1e-2


@pytest.mark.parametrize("n_images", [2, 1, 0])
def test_teacher_forced_decode_logits_every_step(tiny, n_images):
    cfg, W, model = tiny
    imgs = _images(10 + n_images, [(90, 60), (56, 70)][:n_images])
    ids, pv, pm = _request(cfg, imgs, seed=20 + n_images, **SMALL)
    forced = np.random.default_rng(42).integers(3, 1000, 70)
    ref = oi.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pv) if pv is not None else None, pm, forced)
    got, f, n = _engine_teacher_forced(model, ids, pv, pm, forced)
    assert n == ids.shape[1] + len(forced) and got.shape == ref.shape
    worst = _check_rows(got, ref, 2e-2, "idefics2 tiny")
    ok, rep = bf16_close(got, ref, ulps=4, atol_rms=8e-2)
    assert ok, rep
    print(f"idefics2 tiny teacher-forced, {n_images} image(s): worst row rel-rms {worst:.4f}; {rep}")


def test_generate_step_and_batch_generator_multi_image_requests(tiny):
    """generate_step (graph + eager) vs the oracle until a tie; the same requests through the continuous BatchGenerator
    (per-request pixel_values / pixel_attention_mask) produce the tokens they produce alone."""
    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.generate import generate_step

    cfg, W, model = tiny
    reqs = [_request(cfg, _images(30 + i, [(90, 60), (56, 70), (70, 84)][: i % 3 + 1]) if i % 4 else [], seed=40 + i, **SMALL)
            for i in range(6)]
    n_new = 8
    singles = []
    for ids, pv, pm in reqs:
        kw = dict(pixel_attention_mask=pm) if pv is not None else {}
        singles.append([(t, float(lp[t])) for t, lp in generate_step(ids, model, torch.from_numpy(pv) if pv is not None else None,
                                                                     None, max_tokens=n_new, **kw)])
    ids, pv, pm = reqs[1]
    ref_toks, ref_logits = oi.generate_greedy(W, cfg, ids, torch.from_numpy(pv), pm, max_tokens=n_new, return_logits=True)
    for use_graph in (True, False):
        toks = [t for t, _ in generate_step(ids, model, torch.from_numpy(pv), None, max_tokens=n_new, use_graph=use_graph,
                                            pixel_attention_mask=pm)]
        for i in range(n_new):
            if toks[i] != ref_toks[i]:
                r = ref_logits[i].float()
                top2 = r.topk(2).values
                assert float(top2[0] - top2[1]) < 0.06 * float(r.pow(2).mean().sqrt()), (use_graph, i, toks, ref_toks)
                break
        assert toks[0] == ref_toks[0]
    gen = BatchGenerator(model, None, max_tokens=n_new, completion_batch_size=4, prefill_batch_size=4)
    kws = [dict(pixel_values=torch.from_numpy(pv), pixel_attention_mask=pm) if pv is not None else {} for _, pv, pm in reqs]
    uids = gen.insert([r[0].reshape(-1) for r in reqs], [n_new] * len(reqs), prompt_kwargs=kws)
    got = {u: [] for u in uids}
    while gen.has_work:
        _, out = gen.next()
        for r in out:
            got[r.uid].append((r.token, r.token_logprob))
    gen.close()
    # batch rows and single requests run different reduction structures: equal up to bf16 ties (test_engine_gpu.py)
    from tests.test_engine_gpu import _assert_streams_equal_up_to_ties
    _assert_streams_equal_up_to_ties([got[u] for u in uids], singles, min_equal=0.85)


def test_load_from_hf_layout_checkpoint_and_generate(tmp_path):
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from mlx_vlm_amd import utils
    from mlx_vlm_amd.generate import generate_step
    from tests.helpers import idefics2_config_from_oracle

    cfg = oi.tiny_cfg()
    W = oi.random_weights(cfg, seed=9, dtype=BF, **oi.TEST_WEIGHT_SCALES)
    hf = {}
    for k, v in W.items():                                     # HF layout: model.{vision_model,connector,text_model}.*, lm_head.*
        if k.startswith("language_model.lm_head."):
            hf[k[len("language_model."):]] = v.contiguous()
        elif k.startswith("language_model."):
            hf["model.text_model." + k[len("language_model."):]] = v.contiguous()
        else:
            hf["model." + k] = v.contiguous()
    pk = "model.vision_model.embeddings.patch_embedding.weight"
    hf[pk] = hf[pk].permute(0, 3, 1, 2).contiguous()             # torch conv layout
    hf["model.text_model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.zeros(64)
    save_file(hf, str(tmp_path / "model.safetensors"))
    mc = idefics2_config_from_oracle(cfg)
    conf = dict(model_type="idefics2", image_token_id=cfg.image_token_id, vocab_size=cfg.text.vocab_size,
                text_config=mc.text_config.to_dict(), vision_config=mc.vision_config.to_dict(),
                perceiver_config=mc.perceiver_config.to_dict())
    (tmp_path / "config.json").write_text(json.dumps(conf))
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2, **{f"w{i}": i + 3 for i in range(900)}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>")
    fast.add_special_tokens({"additional_special_tokens": ["<fake_token_around_image>", "<image>", "<end_of_utterance>"]})
    fast.save_pretrained(str(tmp_path))
    img_id = fast.convert_tokens_to_ids("<image>")
    conf["image_token_id"] = img_id
    (tmp_path / "config.json").write_text(json.dumps(conf))
    (tmp_path / "preprocessor_config.json").write_text(json.dumps({"size": {"shortest_edge": 56, "longest_edge": 140}}))
    (tmp_path / "processor_config.json").write_text(json.dumps({"image_seq_len": cfg.perceiver.resampler_n_latents}))
    model, proc = utils.load(str(tmp_path), kv_pool_tokens=4096, max_seqs=4)
    assert type(model).__module__.endswith("idefics2.idefics2") and model.config.image_token_index == img_id
    ims = _images(70, [(90, 60), (56, 70)])
    inp = utils.prepare_inputs(proc, images=ims, prompts="w5 w9 <image> w7 <image> w30 w2")
    assert int((inp["input_ids"] == img_id).sum()) == 2 * cfg.perceiver.resampler_n_latents
    cfg2 = oi.tiny_cfg()
    cfg2.image_token_id = img_id
    mem = build_idefics2_model(cfg2, W, kv_pool_tokens=4096, max_seqs=4)
    kw = dict(pixel_attention_mask=inp["pixel_attention_mask"])
    a = [t for t, _ in generate_step(inp["input_ids"], model, torch.from_numpy(inp["pixel_values"]), None, max_tokens=6, **kw)]
    b = [t for t, _ in generate_step(inp["input_ids"], mem, torch.from_numpy(inp["pixel_values"]), None, max_tokens=6, **kw)]
    assert a == b and len(a) == 6


@pytest.mark.parametrize("M", [1, 2, 8])
def test_norm_prologue_gemvs_at_hidden_4096(M):
    """Mistral-7B's hidden size through the row-wave GEMVs with 8 chunks per lane: RMSNorm + gate/up + SwiGLU, RMSNorm + head,
    RMSNorm + qkv + RoPE + paged KV write - vs the oracle (2 ulps: fp32 accumulation in another order)."""
    from mlx_vlm_amd import ops as vops

    K, I = 4096, 1024
    g = torch.Generator().manual_seed(5 + M)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(BF)   # noqa: E731
    h, nw = rnd(M, K), (1 + 0.1 * torch.randn(K, generator=g)).to(BF)
    wg, wu = rnd(I, K, scale=0.03), rnd(I, K, scale=0.03)
    wgu = torch.stack([wg, wu], 1).reshape(2 * I, K)
    xn = O.rms_norm(h, nw, 1e-5)
    out = vops.gemv(h.cuda(), wgu.cuda(), norm_w=nw.cuda(), eps=1e-5, epilogue=vops.EPI_SWIGLU)
    ok, rep = bf16_close(out, O.swiglu(O.linear(xn, wg), O.linear(xn, wu)), ulps=3)
    assert ok, rep
    wh = rnd(1003, K, scale=0.03)                                  # ragged N
    ok, rep = bf16_close(vops.gemv(h.cuda(), wh.cuda(), norm_w=nw.cuda(), eps=1e-5), O.linear(xn, wh), ulps=2)
    assert ok, rep
    Hq, Hkv, D = 4, 1, 128
    wqkv, bqkv = rnd((Hq + 2 * Hkv) * D, K, scale=0.03), torch.zeros((Hq + 2 * Hkv) * D, dtype=BF)
    pos = torch.randint(0, 3000, (M,), generator=g, dtype=torch.int32)
    max_pages = 4
    slot = torch.randint(0, 64 * max_pages, (M,), generator=g, dtype=torch.int32)
    inv = O.mrope_inv_freq(D, 1e4)
    qkv = O.linear(xn, wqkv).view(M, Hq + 2 * Hkv, D)
    p2 = pos.long()[:, None]
    qr = O.mrope_apply(qkv[:, :Hq][:, :, None], p2, inv, None, "fused")[:, :, 0]
    n_pages = M * max_pages
    bt = torch.randperm(n_pages, generator=g).to(torch.int32).reshape(M, max_pages)
    kpool = torch.zeros(n_pages, Hkv, D // 8, 64, 8, dtype=BF, device="cuda")
    vpool = torch.zeros(n_pages, Hkv, D, 64, dtype=BF, device="cuda")
    out = vops.gemv_qkv_rope_kvwrite(h.cuda(), nw.cuda(), wqkv.cuda(), bqkv.cuda(), Hq, Hkv, D, pos.cuda(), slot.cuda(), inv.cuda(),
                                     bt.cuda(), kpool, vpool, eps=1e-5)
    ok, rep = bf16_close(out.view(M, Hq + 2 * Hkv, D)[:, :Hq], qr, ulps=2)
    assert ok, rep


def test_idefics2_8b_widths_four_images_vs_oracle():
    """BASELINE configs[3] at Idefics2-8B's widths and reduced depth: SigLIP-so400m tower (1152 / 4304, 16 heads of 72, 2
    layers), perceiver (64 latents, 16 heads of 96 over 4 kv heads, 3 layers), Mistral-7B decoder (4096 / 14336, 32 heads over
    8 kv heads, 2 layers), vocabulary 32003; one prompt with 4 x 336 x 336 images (378 x 378 after the resize rule: 729 patches
    each -> 4 x 64 image tokens) + text: resampler outputs, prefill logits and 6 teacher-forced decode steps."""
    cfg = oi.Cfg(text=oi.TextCfg(num_hidden_layers=2), vision=oi.VisionCfg(num_hidden_layers=2), perceiver=oi.PerceiverCfg())
    assert cfg.text.vocab_size == 32003 and cfg.vision.num_patches_per_side == 70
    W = oi.random_weights(cfg, seed=17, dtype=BF, std=0.02, embed_std=0.02)
    model = build_idefics2_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    imgs = _images(80, [(336, 336)] * 4)
    ids, pv, pm = _request(cfg, imgs, seed=81, n_text=(20, 30, 30), vocab_hi=32000)
    assert pv.shape == (1, 4, 3, 378, 378) and int((ids == cfg.image_token_id).sum()) == 256
    ref_feats = oi.image_features(W, cfg, torch.from_numpy(pv).to(BF), pm)
    got_feats = model.encode_image(torch.from_numpy(pv), pm)
    e = _rel_rms(got_feats, ref_feats.reshape(-1, ref_feats.shape[-1]))
    assert e < 2e-2, e
    forced = np.random.default_rng(82).integers(3, 32000, 6)
    ref = oi.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pv), pm, forced)
    got, _, n = _engine_teacher_forced(model, ids, pv, pm, forced)
    assert n == ids.shape[1] + 6 and got.shape == ref.shape == (7, 32003)
    worst = _check_rows(got, ref, 2e-2, "idefics2-8b widths")
    print(f"idefics2-8b widths: feature rel-rms {e:.4f}, worst logit-row rel-rms {worst:.4f}")
