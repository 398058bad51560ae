"""GPU parity at the sizes the benchmarks run ("next" item 1): BASELINE configs[2], [3] and [4] at FULL
depth, and the 16-row decode step against the oracle.

  * configs[2]  Qwen2-VL-7B: 32 ViT blocks of 1280 (merger out 3584), 28 decoder layers of 3584 / 18944, GQA 28:4,
                untied lm_head, V = 152,064; one 336 x 336 image (144 image tokens) + 128 text tokens;
  * configs[3]  Idefics2-8B: SigLIP-so400m (27 layers of 1152 / 4304), perceiver resampler (3 layers, 64 latents),
                Mistral-7B (32 layers of 4096 / 14336, GQA 32:8), V = 32,003; one prompt with 4 x 336 x 336 images;
  * configs[4]  Phi-3.5-vision: CLIP ViT-L/14-336 (24 layers, 23 run), HD transform, Phi-3 decoder (32 layers of
                3072 / 8192, 32 heads of 96) with an MLX 4-bit language model, V = 32,064; one 336 x 336 image;
each: image features, last-row prefill logits and 4 teacher-forced decode steps (a seeded random token stream, every
step's logits) against the oracle on the same synthetic checkpoint - the form of
tests/test_parity_decode_gpu.py::test_full_depth_qwen2_vl_2b_image_prefill_and_teacher_forced_decode;
  * 16 rows (and 9) through vlm_llm_decode_forward - every projection on the skinny-M MFMA GEMM, qkv + rope + KV write
    in its epilogue - with EVERY row's logits compared with the oracle's (not with single-request runs of the engine),
    bf16 (Qwen2-VL) and MLX 4-bit (Phi-3).

Tolerances per depth (rel-rms of a logit row against the oracle; stated at each assert): the engine and the oracle
round at the same points but accumulate in different orders, a 1-ulp flip of a bf16 activation is amplified by the
layers after it; measured distances grow as ~sqrt(depth): 2 layers 0.3-1.2 %, 28 layers 2-3 %, 32 layers + a 4-bit
language model 3-4 %.

The synthetic checkpoints use oracle.ops.fast_normal for their big matrices (windows of one normal pool: seconds instead
of minutes per 8 B-parameter model on the GPU box's clock); the oracle and the engine read the same tensors.
"""
import numpy as np
import pytest
import torch

from tests.helpers import build_idefics2_model, build_phi3v_model, build_product_model

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
N_FORCED = 4


def _rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


def _check_rows(got, ref, tol_rms, tag):
    """every row within tol_rms (rel-rms); argmax identical wherever the oracle's top-2 margin exceeds 0.25 rms (far
    above the per-element error at every tolerance used here)"""
    errs = [_rel_rms(got[i], ref[i]) for i in range(ref.shape[0])]
    print(f"{tag}: row rel-rms " + " ".join(f"{e:.4f}" for e in errs))
    assert max(errs) < tol_rms, (tag, errs)
    for i in range(ref.shape[0]):
        r = ref[i].float()
        top2 = r.topk(2).values
        if float(top2[0] - top2[1]) > 0.25 * float(r.pow(2).mean().sqrt()):
            assert int(got[i].float().argmax()) == int(r.argmax()), (tag, i)
    return max(errs)


def _teacher_forced(model, ids, pixels, forced, **kw):
    lm = model.language_model
    f = model.get_input_embeddings(ids, pixels, **kw)
    cache = lm.make_cache()
    out = lm(ids, f.inputs_embeds, cache=cache, position_ids=f.position_ids, rope_deltas=f.rope_deltas, logits_to_keep=1)
    rows = [out.logits[0, -1].clone()]
    for y in forced:
        rows.append(lm(np.array([[int(y)]]), cache=cache).logits[0, -1].clone())
    n = cache[0].offset
    cache[0]._seq.release()
    return torch.stack(rows), n


# ------------------------------------------------------------------------------------------------ BASELINE configs[2]
def test_full_depth_qwen2_vl_7b_image_prefill_and_teacher_forced_decode():
    from oracle import image_processor as oip
    from oracle import qwen2_vl as oq

    text = oq.TextCfg(hidden_size=3584, num_hidden_layers=28, intermediate_size=18944, num_attention_heads=28,
                      num_key_value_heads=4, vocab_size=152064, tie_word_embeddings=False)
    cfg = oq.Cfg(text=text, vision=oq.VisionCfg(depth=32, embed_dim=1280, hidden_size=3584, num_heads=16))
    W = oq.random_weights(cfg, seed=70, dtype=BF, std=0.02, fast=True)
    model = build_product_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    img = np.random.default_rng(31).integers(0, 256, (3, 336, 336), dtype=np.uint8)
    pix, thw = oip.process([img])
    n_img = int(thw.prod()) // 4
    assert (pix.shape[0], n_img) == (576, 144)
    text_ids = np.random.default_rng(1031).integers(0, 151643, 128)
    ids = np.concatenate([[cfg.vision_start_token_id], np.full(n_img, cfg.image_token_id), [cfg.vision_start_token_id + 1],
                          text_ids]).astype(np.int64)[None]
    forced = np.random.default_rng(1032).integers(0, 151643, N_FORCED)
    pix_t = torch.from_numpy(pix).to(BF)
    ref_feats = oq.vision_tower(W, cfg, pix_t, thw)
    ref = oq.decode_teacher_forced(W, cfg, ids, pix_t, thw, forced)
    feats = model.vision_tower(torch.from_numpy(pix), thw)
    e_feat = _rel_rms(feats, ref_feats)
    assert e_feat < 3e-2, e_feat                                    # 32 blocks + merger in bf16
    got, n = _teacher_forced(model, ids, torch.from_numpy(pix), forced, image_grid_thw=thw)
    assert n == ids.shape[1] + N_FORCED == 274 + N_FORCED and got.shape == ref.shape == (1 + N_FORCED, 152064)
    worst = _check_rows(got, ref, 6e-2, "7B")                       # 28 layers of bf16 after a 32-block tower
    print(f"full-depth 7B: feature rel-rms {e_feat:.4f}, worst logit-row rel-rms {worst:.4f}")


# ------------------------------------------------------------------------------------------------ BASELINE configs[3]
def test_full_depth_idefics2_8b_four_images_prefill_and_teacher_forced_decode():
    from oracle import idefics2 as oi
    from tests.test_vlm_family_idefics2_gpu import _images, _request

    cfg = oi.Cfg(text=oi.TextCfg(), vision=oi.VisionCfg(), perceiver=oi.PerceiverCfg())
    assert (cfg.text.num_hidden_layers, cfg.vision.num_hidden_layers, cfg.text.vocab_size) == (32, 27, 32003)
    W = oi.random_weights(cfg, seed=71, dtype=BF, std=0.02, embed_std=0.02, fast=True)
    model = build_idefics2_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    imgs = _images(90, [(336, 336)] * 4)
    ids, pv, pm = _request(cfg, imgs, seed=91, n_text=(20, 30, 30), vocab_hi=32000)
    assert pv.shape == (1, 4, 3, 378, 378) and int((ids == cfg.image_token_id).sum()) == 256
    ref_feats = oi.image_features(W, cfg, torch.from_numpy(pv).to(BF), pm)
    got_feats = model.encode_image(torch.from_numpy(pv), pm)
    e_feat = _rel_rms(got_feats, ref_feats.reshape(-1, ref_feats.shape[-1]))
    assert e_feat < 3e-2, e_feat                                    # 27 SigLIP layers + 3 perceiver layers in bf16
    forced = np.random.default_rng(92).integers(3, 32000, N_FORCED)
    ref = oi.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pv), pm, forced)
    got, n = _teacher_forced(model, ids, torch.from_numpy(pv), forced, pixel_attention_mask=pm)
    assert n == ids.shape[1] + N_FORCED and got.shape == ref.shape == (1 + N_FORCED, 32003)
    worst = _check_rows(got, ref, 6e-2, "idefics2-8b")              # 32 layers of bf16 behind tower + resampler
    print(f"full-depth Idefics2-8B: feature rel-rms {e_feat:.4f}, worst logit-row rel-rms {worst:.4f}")


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]
def test_full_depth_phi35_vision_4bit_prefill_and_teacher_forced_decode():
    from oracle import phi3_v as op
    from oracle import quant as Q
    from tests.test_vlm_family_phi3v_gpu import _images, _request

    short, long = op.su_factors(96, seed=9)
    cfg = op.Cfg(text=op.TextCfg(short_factor=short, long_factor=long), vision=op.VisionCfg())
    assert (cfg.text.num_hidden_layers, cfg.vision.num_hidden_layers, cfg.text.vocab_size) == (32, 24, 32064)
    W = op.random_weights(cfg, seed=72, dtype=BF, std=0.02, embed_std=0.02, fast=True)
    # MLX affine 4-bit language model (group 64), bf16 tower / projection: ck = the packed checkpoint the engine loads,
    # ow = the oracle's view of the same quantized weights
    ck, ow = Q.quantize_checkpoint(W, predicate=lambda p, v: not p.startswith("model.vision_embed_tokens."))
    del W
    model = build_phi3v_model(cfg, ck, kv_pool_tokens=4096, max_seqs=4)
    imgs = _images(93, [(336, 336)])
    ids, pv, sz = _request(cfg, imgs, n_text=(20, 40, 40), seed=94, vocab_hi=32000)
    assert int((ids < 0).sum()) == 757
    forced = np.random.default_rng(95).integers(3, 32000, N_FORCED)
    ref_rows = op.image_features(ow, cfg, torch.from_numpy(pv).to(BF), sz)[0]
    got_rows = model.vision_model.image_features(torch.from_numpy(pv), sz)[0]
    e_feat = _rel_rms(got_rows, ref_rows)
    assert e_feat < 3e-2, e_feat                                    # 23 CLIP layers + HD projection in bf16
    ref = op.decode_teacher_forced(ow, cfg, ids, torch.from_numpy(pv), sz, forced)
    got, n = _teacher_forced(model, ids, torch.from_numpy(pv), forced, image_sizes=sz)
    assert n == ids.shape[1] + N_FORCED and got.shape == ref.shape == (1 + N_FORCED, 32064)
    worst = _check_rows(got, ref, 7e-2, "phi3.5-vision 4-bit")      # 32 layers, 4-bit weights amplify activation flips
    print(f"full-depth Phi-3.5-vision (4-bit LM): feature rel-rms {e_feat:.4f}, worst logit-row rel-rms {worst:.4f}")


# ------------------------------------------------------------------------------------------------ 16 decode rows vs oracle
def _rows_vs_oracle(model, lm_decode_ref, prompts, forced, tol, tag):
    """prompts: B token arrays; forced: [steps][B].  Prefill each prompt into its own cache (consecutive KV slots), then
    every step feeds one token per row through ONE B-row vlm_llm_decode_forward; every row of every step vs the
    oracle's teacher-forced logits of that sequence."""
    lm = model.language_model
    B = len(prompts)
    caches = []
    for p in prompts:
        c = lm.make_cache()
        lm(p[None], cache=c, logits_to_keep=1)
        caches.append(c)
    got = []
    for step in forced:
        got.append(lm(np.asarray(step, dtype=np.int64).reshape(B, 1), cache=caches).logits[:, 0].clone())
    got = torch.stack(got)                                          # [steps, B, V]
    for c in caches:
        c[0]._seq.release()
    worst = 0.0
    for b in range(B):
        ref = lm_decode_ref(prompts[b], [int(s[b]) for s in forced])[1:]      # drop the prefill row
        worst = max(worst, _check_rows(got[:, b], ref, tol, f"{tag} row {b}"))
    return worst


@pytest.mark.parametrize("B", [16, 9, 32, 57])
def test_decode_forward_16_rows_every_row_vs_oracle_bf16(B):
    """Qwen2-VL tiny dims, B rows of different context lengths (3 .. 140 tokens: rows on their first page, rows past a page
    boundary) through the B-row decode forward (B = 9 runs as its own width: rows as the MFMA N dimension, 7 idle
    columns; B = 32 / 57: WIDE steps - the prefill GEMMs, M-RoPE + KV write at the rows' slots, paged decode attention):
    4 steps, every row's logits vs the oracle; 2 layers of bf16: 2e-2."""
    from oracle import qwen2_vl as oq

    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    model = build_product_model(cfg, W, kv_pool_tokens=32768, max_seqs=72)
    rng = np.random.default_rng(500 + B)
    prompts = [rng.integers(3, 1000, 3 + (b * 37) % 138).astype(np.int64) for b in range(B)]
    forced = rng.integers(3, 1000, (4, B))
    worst = _rows_vs_oracle(model, lambda p, f: oq.decode_teacher_forced(W, cfg, p[None], None, None, np.asarray(f)),
                            prompts, forced, 2e-2, f"bf16 B={B}")
    print(f"{B}-row decode forward vs oracle (bf16): worst row rel-rms {worst:.4f}")


def test_decode_forward_16_rows_every_row_vs_oracle_4bit():
    """the same over an MLX 4-bit Phi-3 tiny model (dequant-fused MFMA form of the skinny-M GEMM, Su-RoPE in the qkv
    epilogue): 16 rows, 4 steps, every row vs the oracle's 4-bit graph; 2e-2."""
    from oracle import phi3_v as op
    from oracle import quant as Q

    cfg = op.tiny_cfg()
    W = op.random_weights(cfg, seed=4321, dtype=BF, std=0.04, embed_std=0.2)
    ck, ow = Q.quantize_checkpoint(W, predicate=lambda p, v: not p.startswith("model.vision_embed_tokens."))
    model = build_phi3v_model(cfg, ck, kv_pool_tokens=16384, max_seqs=32)
    rng = np.random.default_rng(516)
    B = 16
    prompts = [rng.integers(3, 1000, 3 + (b * 37) % 138).astype(np.int64) for b in range(B)]
    forced = rng.integers(3, 1000, (4, B))
    worst = _rows_vs_oracle(model, lambda p, f: op.decode_teacher_forced(ow, cfg, p[None], None, None, np.asarray(f)),
                            prompts, forced, 2e-2, "4-bit B=16")
    print(f"16-row decode forward vs oracle (4-bit): worst row rel-rms {worst:.4f}")


# ------------------------------------------------------------------------------------------------ the benchmarked WIDE shapes
def test_wide_32_rows_at_7b_widths_every_row_checked():
    """The configuration behind the 7B batch-32 line: 32 rows through ONE vlm_llm_decode_forward at
    Qwen2-VL-7B WIDTHS - hidden 3584, inter 18944, GQA 28:4, V = 152,064 (2 layers) - i.e. the tile / split shapes the tiny
    wide tests never reach: gemm256 on gate/up, split-K 64x64 GEMMs with the shared capture-stream workspace on o_proj /
    down, splitk_reduce, the 152,064-column head GEMM.  One row sits past 2048 tokens of context, so the step runs the
    split (nsplit > 1) attention whose merged output the wide o_proj GEMM must read (it read the RMSNorm
    scratch).  EVERY row of 3 steps is compared with the same row decoded ALONE through the one-row engine path (itself
    pinned to the oracle at full depth above; 2e-2 rel-rms - different reduction orders), and 7 rows spread over the tile
    (first / last row of the two 16-row halves, the long row, a page-boundary row) with the ORACLE's teacher-forced logits."""
    from oracle import qwen2_vl as oq

    text = oq.TextCfg(hidden_size=3584, num_hidden_layers=2, intermediate_size=18944, num_attention_heads=28,
                      num_key_value_heads=4, vocab_size=152064, tie_word_embeddings=False)
    cfg = oq.Cfg(text=text, vision=oq.VisionCfg(depth=1, embed_dim=1280, hidden_size=3584, num_heads=16))
    W = oq.random_weights(cfg, seed=73, dtype=BF, std=0.02, fast=True)
    model = build_product_model(cfg, W, kv_pool_tokens=32768, max_seqs=72)
    lm = model.language_model
    B, steps = 32, 3
    rng = np.random.default_rng(730)
    lens = [5 + (b * 41) % 190 for b in range(B)]
    lens[7], lens[20] = 2100, 64                       # beyond 2048 tokens: split attention; exactly one full page
    prompts = [rng.integers(0, 151643, n).astype(np.int64) for n in lens]
    forced = rng.integers(0, 151643, (steps, B))
    caches = []
    for p in prompts:
        c = lm.make_cache()
        lm(p[None], cache=c, logits_to_keep=1)
        caches.append(c)
    got = torch.stack([lm(forced[s].reshape(B, 1), cache=caches).logits[:, 0].clone() for s in range(steps)])   # [steps, B, V]
    for c in caches:
        c[0]._seq.release()
    assert got.shape == (steps, B, 152064)
    worst_single = 0.0
    for b in range(B):                                  # every row vs the one-row path
        c = lm.make_cache()
        lm(prompts[b][None], cache=c, logits_to_keep=1)
        for s in range(steps):
            alone = lm(np.array([[int(forced[s, b])]]), cache=c).logits[0, 0]
            e = _rel_rms(got[s, b], alone)
            worst_single = max(worst_single, e)
            assert e < 2e-2, ("row alone", b, s, e)
        c[0]._seq.release()
    worst = 0.0
    for b in (0, 7, 15, 16, 20, 23, 31):                # ... and these against the oracle
        ref = oq.decode_teacher_forced(W, cfg, prompts[b][None], None, None, forced[:, b])[1:]
        worst = max(worst, _check_rows(got[:, b], ref, 2e-2, f"7B-width wide row {b}"))
    print(f"32 wide rows at 7B widths: worst vs one-row path {worst_single:.4f}, worst vs oracle {worst:.4f}")


_TAILS_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from tests.test_full_depth_gpu import _wide_tail_logits
got, launches = _wide_tail_logits(sys.argv[3])
np.savez(sys.argv[2], logits=got.view(__import__("torch").int16).cpu().numpy(), launches=launches)
"""


def _wide_tail_logits(widths):
    """32 rows x 2 wide steps at 7B / 2B widths (2 layers; one row on split attention) -> (logits [steps, B, V], launches of a step)"""
    from mlx_vlm_amd import _lib
    from oracle import qwen2_vl as oq

    H, I, NH, NKV, V = {"7b": (3584, 18944, 28, 4, 152064), "2b": (1536, 8960, 12, 2, 151936)}[widths]
    text = oq.TextCfg(hidden_size=H, num_hidden_layers=2, intermediate_size=I, num_attention_heads=NH,
                      num_key_value_heads=NKV, vocab_size=V, tie_word_embeddings=False)
    cfg = oq.Cfg(text=text, vision=oq.VisionCfg(depth=1, embed_dim=1280, hidden_size=H, num_heads=16))
    W = oq.random_weights(cfg, seed=74, dtype=BF, std=0.02, fast=True)
    model = build_product_model(cfg, W, kv_pool_tokens=32768, max_seqs=72)
    lm = model.language_model
    B, steps = 32, 2
    rng = np.random.default_rng(731)
    lens = [5 + (b * 43) % 150 for b in range(B)]
    lens[9] = 2100
    prompts = [rng.integers(0, 151643, n).astype(np.int64) for n in lens]
    forced = rng.integers(0, 151643, (steps, B))
    caches = []
    for p in prompts:
        c = lm.make_cache()
        lm(p[None], cache=c, logits_to_keep=1)
        caches.append(c)
    got = torch.stack([lm(forced[s].reshape(B, 1), cache=caches).logits[:, 0].clone() for s in range(steps)])
    torch.cuda.synchronize()
    launches = int(_lib.lib().vlm_llm_decode_launches(lm._handle))
    for c in caches:
        c[0]._seq.release()
    return got, launches


@pytest.mark.parametrize("widths", ["7b", "2b"])
def test_wide_step_reduce_tails_equal_the_separate_launches(tmp_path, widths):
    """Round 6: in a wide step the reduce launch of each split-K GEMM also does its follower - M-RoPE + KV write after qkv, RMSNorm
    after o_proj and after down (csrc/gemm_bf16.hip splitk_reduce_rope_kernel / splitk_reduce_norm_kernel).  Same arithmetic in
    the same order, so the logits of 32 rows x 2 steps at 7B widths (the second step reads the K / V the first one's fused launch
    wrote) equal the separate-launch sequence (VLM_WIDE_TAILS=0, a child process: the knob is read once) BIT FOR BIT, in 13
    launches per step instead of 19.  At 2B widths (K = 1536) the qkv / o_proj GEMMs are split-K only under the <= 64-row policy."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_path = str(tmp_path / "separate.npz")
    r = subprocess.run([sys.executable, "-c", _TAILS_CHILD, root, ref_path, widths], env=dict(os.environ, VLM_WIDE_TAILS="0"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(ref_path)
    got, launches = _wide_tail_logits(widths)
    assert int(ref["launches"]) == 19 and launches == 13, (int(ref["launches"]), launches)
    g = got.view(torch.int16).cpu().numpy()
    assert np.array_equal(g, ref["logits"]), int((g != ref["logits"]).sum())


@pytest.mark.parametrize("B", [20, 40])
def test_wide_rows_over_the_quantized_kv_cache_every_row_vs_oracle(B):
    """Wide steps x kv_bits = 8: B rows whose caches were quantised after their prefill
    (KVCache.to_quantized) decode through wide steps - bf16 K / V of the step's token written by vlm_mrope_kvwrite_decode,
    quantised by the attention launch, scores over the u8 pools - every row of 4 steps against the oracle's quantised
    graph (QuantizedKVCache + quantized SDPA, pinned to the reference's own files in test_oracle_ref_golden_kvquant.py)."""
    from oracle import qwen2_vl as oq

    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    model = build_product_model(cfg, W, kv_pool_tokens=32768, max_seqs=72)
    lm = model.language_model
    rng = np.random.default_rng(740 + B)
    prompts = [rng.integers(3, 1000, 3 + (b * 37) % 138).astype(np.int64) for b in range(B)]
    forced = rng.integers(3, 1000, (4, B))
    caches = []
    for p in prompts:
        c = lm.make_cache()
        lm(p[None], cache=c, logits_to_keep=1)
        caches.append(c)
    lm.quantize_kv([c[0]._seq for c in caches], bits=8, group_size=64)
    got = torch.stack([lm(forced[s].reshape(B, 1), cache=caches).logits[:, 0].clone() for s in range(4)])
    for c in caches:
        c[0]._seq.release()
    worst, worst_plain = 0.0, 0.0
    for b in range(B):
        ref = oq.decode_teacher_forced(W, cfg, prompts[b][None], None, None, forced[:, b], kv_bits=8, quantized_kv_start=0)[1:]
        worst = max(worst, _check_rows(got[:, b], ref, 2e-2, f"wide q8 B={B} row {b}"))
    # the steps really ran over the 8-bit pools: a long row follows the quantised oracle more closely than the bf16 one
    b = max(range(B), key=lambda i: len(prompts[i]))
    ref_q = oq.decode_teacher_forced(W, cfg, prompts[b][None], None, None, forced[:, b], kv_bits=8, quantized_kv_start=0)[1:]
    ref_p = oq.decode_teacher_forced(W, cfg, prompts[b][None], None, None, forced[:, b])[1:]
    dq = np.mean([_rel_rms(got[s, b], ref_q[s]) for s in range(4)])
    dp = np.mean([_rel_rms(got[s, b], ref_p[s]) for s in range(4)])
    assert dq < dp, (dq, dp)
    print(f"{B} wide rows over 8-bit K/V vs oracle: worst row rel-rms {worst:.4f} (to the quantised graph {dq:.4f}, to bf16 {dp:.4f})")


# ------------------------------------------------------------------------------------------------ greedy token identity at full depth
def _greedy_identity(tag, model, oracle_greedy, ids, n_new, k_oracle, stride, n_cycle):
    """32 greedy tokens through the captured decode step must be IDENTICAL to the oracle's - no tie rule (north_star: "token-id
    bit-exact under greedy").  The oracle decodes the first k_oracle tokens at full depth (seconds per token on the host)
    and must itself show the construction's margin at every one of them (top-2 gap above half the logit rms) and walk the
    successor permutation; the engine's whole stream must equal the oracle's prefix and continue the same permutation walk,
    which the construction makes the unique greedy path."""
    from mlx_vlm_amd.generate import generate_step

    ref_toks, ref_logits = oracle_greedy(k_oracle)
    want = [(int(ids[0, -1]) + stride * (i + 1)) % n_cycle for i in range(n_new)]
    assert ref_toks == want[:k_oracle], (tag, ref_toks, want[:k_oracle])
    for i in range(k_oracle):
        r = ref_logits[i].float()
        top2 = r.topk(2).values
        assert float(top2[0] - top2[1]) > 0.5 * float(r.pow(2).mean().sqrt()), (tag, i)
    toks = [t for t, _ in generate_step(ids, model, None, None, max_tokens=n_new, temperature=0.0)]
    assert toks[:k_oracle] == ref_toks, (tag, toks[:k_oracle], ref_toks)
    assert toks == want, (tag, toks, want)
    print(f"{tag}: {n_new} greedy tokens identical ({k_oracle} against the oracle, the rest on the permutation it confirmed)")


def test_full_depth_qwen2_vl_7b_greedy_tokens_identical_to_oracle():
    from oracle import qwen2_vl as oq
    from tests.helpers import peaked_full_depth

    text = oq.TextCfg(hidden_size=3584, num_hidden_layers=28, intermediate_size=18944, num_attention_heads=28,
                      num_key_value_heads=4, vocab_size=152064, tie_word_embeddings=False)
    cfg = oq.Cfg(text=text, vision=oq.VisionCfg(depth=1, embed_dim=1280, hidden_size=3584, num_heads=16))
    W = oq.random_weights(cfg, seed=74, dtype=BF, std=0.02, fast=True)
    W = peaked_full_depth(W, "language_model.model.embed_tokens.weight", "language_model.lm_head.weight", gamma=0.2, n_cycle=150000)
    model = build_product_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    ids = np.random.default_rng(741).integers(3, 150000, (1, 40))
    _greedy_identity("7B", model, lambda k: oq.generate_greedy(W, cfg, ids, max_tokens=k, return_logits=True), ids, 32, 8, 389, 150000)


def test_full_depth_idefics2_8b_greedy_tokens_identical_to_oracle():
    from oracle import idefics2 as oi
    from tests.helpers import peaked_full_depth

    cfg = oi.Cfg(text=oi.TextCfg(), vision=oi.VisionCfg(num_hidden_layers=1), perceiver=oi.PerceiverCfg())
    W = oi.random_weights(cfg, seed=75, dtype=BF, std=0.02, embed_std=0.02, fast=True)
    W = peaked_full_depth(W, oi.LM + "embed_tokens.weight", oi.LM + "lm_head.weight", gamma=0.2, n_cycle=32000)
    model = build_idefics2_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    ids = np.random.default_rng(751).integers(3, 32000, (1, 40))
    _greedy_identity("Idefics2-8B", model, lambda k: oi.generate_greedy(W, cfg, ids, max_tokens=k, return_logits=True), ids, 32, 8,
                     389, 32000)


def test_full_depth_phi35_vision_4bit_greedy_tokens_identical_to_oracle():
    from oracle import phi3_v as op
    from oracle import quant as Q
    from tests.helpers import peaked_full_depth

    short, long = op.su_factors(96, seed=9)
    cfg = op.Cfg(text=op.TextCfg(short_factor=short, long_factor=long), vision=op.VisionCfg(num_hidden_layers=2))
    W = op.random_weights(cfg, seed=76, dtype=BF, std=0.02, embed_std=0.02, fast=True)
    W = peaked_full_depth(W, op.M + "embed_tokens.weight", "lm_head.weight", gamma=0.25, n_cycle=32000)
    ck, ow = Q.quantize_checkpoint(W, predicate=lambda p, v: not p.startswith("model.vision_embed_tokens."))
    del W
    model = build_phi3v_model(cfg, ck, kv_pool_tokens=4096, max_seqs=4)
    ids = np.random.default_rng(761).integers(3, 32000, (1, 40))
    Q.QW.keep_f32 = True                 # (5 full-size oracle tokens: unpack each 4-bit matrix once, not per call)
    try:
        _greedy_identity("Phi-3.5-vision 4-bit", model, lambda k: op.generate_greedy(ow, cfg, ids, max_tokens=k, return_logits=True),
                         ids, 32, 5, 389, 32000)
    finally:
        Q.QW.keep_f32 = False
