"""Pin the oracle's QUANTISED paths to vectors produced by the REFERENCE'S OWN files (tests/golden/kvquant_ref.npz, made by
tests/golden/make_golden_ref_kvquant.py: models/cache.py QuantizedKVCache / to_quantized / should_quantize_kv_layer,
models/base.py quantized_scaled_dot_product_attention, generate/common.py maybe_quantize_kv_cache, generate/ar.py
generate_step(kv_bits=8), utils.py load_model on an MLX 4-bit checkpoint - all imported unmodified from /root/reference and
run over oracle/mlx_shim).

What this pins, bit for bit: the GRAPH around mx.quantize / mx.quantized_matmul / mx.dequantize - when each layer's cache
switches, what is quantised (new rows on update, the whole cache on to_quantized), the typed `queries *= scale`, the GQA
expansion, mask + softmax(precise) in the score dtype, one rounding per quantized matmul; which modules nn.quantize swaps
(the predicate of utils.py:918-967), QuantizedEmbedding rows and the tied head through `as_linear`.  What it cannot pin: the
arithmetic inside those three MLX primitives - the shim calls oracle/quant.py's single restatement of MLX's published
algorithm for them (stated in both headers; "parity unpinned" for that arithmetic only).
CPU only; nothing here reads /root/reference.
"""
import os

import numpy as np
import torch

from oracle import ops, quant
from oracle import qwen2_vl as oq

R = np.load(os.path.join(os.path.dirname(__file__), "golden", "kvquant_ref.npz"))
BF = torch.bfloat16


def bf(a):
    return torch.from_numpy(np.asarray(a)).to(BF)


def words(t):
    return t.contiguous().numpy().view(np.uint32)


def same_tuple(got, prefix):
    assert np.array_equal(words(got[0]), R[prefix + ".words"]), prefix
    assert np.array_equal(got[1].float().numpy(), R[prefix + ".scales"]), prefix
    assert np.array_equal(got[2].float().numpy(), R[prefix + ".biases"]), prefix


def test_quantized_kv_cache_update_and_to_quantized_match_reference():
    k1, v1, k2, v2 = (bf(R["op." + n]) for n in ("k1", "v1", "k2", "v2"))
    qc = quant.QuantizedKVCache(64, 8)
    qc.update_and_fetch(k1, v1)
    K, V = qc.update_and_fetch(k2, v2)
    assert qc.offset == 6
    same_tuple(K, "op.qcache.keys")
    same_tuple(V, "op.qcache.values")
    kc = ops.KVCache()
    kc.update_and_fetch(k1, v1)
    kc.update_and_fetch(k2, v2)
    tq = quant.to_quantized(kc, 64, 8)
    assert tq.offset == 6
    same_tuple(tq.keys, "op.to_quantized.keys")
    same_tuple(tq.values, "op.to_quantized.values")
    # quantise-on-update == quantise-the-whole-cache (groups never straddle tokens): the reference's two routes agree
    assert np.array_equal(R["op.qcache.keys.words"], R["op.to_quantized.keys.words"])
    # the special groups of the fixture: an all-zero group keeps scale = the 1e-7 floor rounded to bf16 and bias 0
    assert R["op.qcache.keys.biases"][0, 0, 2, 0] == 0.0 and abs(R["op.qcache.keys.scales"][0, 0, 2, 0]) < 2e-7


def test_quantized_sdpa_matches_reference():
    k1, v1, k2, v2 = (bf(R["op." + n]) for n in ("k1", "v1", "k2", "v2"))
    qc = quant.QuantizedKVCache(64, 8)
    qc.update_and_fetch(k1, v1)
    K, V = qc.update_and_fetch(k2, v2)
    od = quant.quantized_sdpa(bf(R["op.q_decode"]), K, V, scale=128 ** -0.5, causal=False)
    om = quant.quantized_sdpa(bf(R["op.q_multi"]), K, V, scale=128 ** -0.5, causal=True)
    assert np.array_equal(od.float().numpy(), R["op.sdpa_decode"])
    assert np.array_equal(om.float().numpy(), R["op.sdpa_causal"])


def test_batch_layer_policy_matches_reference():
    for n, i, want in R["policy.should_quantize_kv_layer"].tolist():
        assert quant.should_quantize_kv_layer(i, n) == bool(want), (n, i)


def peaked():
    cfg = oq.tiny_cfg()
    cfg.text.tie_word_embeddings = False
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    for k in list(W):
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            W[k] = (W[k].float() * 0.5).to(BF)
    return cfg, oq.peak_head(W, cfg, gamma=4.0, stride=389, n_cycle=1000)


def test_generate_step_kv_bits_tokens_and_logprobs_match_reference():
    """generate_step(kv_bits=8, kv_group_size=64, quantized_kv_start=s): s = 0 (quantised right after the prefill), 24 (the
    19-token prompt's cache switches after the 5th decode forward), 10**6 (never): tokens and every bf16 log-prob row."""
    cfg, W = peaked()
    ids = R["gen.text.input_ids"]
    for tag, start in (("s0", 0), ("s24", 24), ("never", 10 ** 6)):
        toks, lp = oq.generate_greedy(W, cfg, ids, max_tokens=12, rope_mode="fallback", kv_bits=8, kv_group_size=64,
                                      quantized_kv_start=start, return_logprobs=True)
        assert toks == R[f"gen.text.{tag}.tokens"].tolist(), tag
        assert np.array_equal(lp.float().numpy(), R[f"gen.text.{tag}.logprobs"]), tag
    # the three policies are different computations (the pin can tell them apart)
    assert not np.array_equal(R["gen.text.s0.logprobs"], R["gen.text.never.logprobs"])
    assert np.array_equal(R["gen.text.s24.logprobs"][:5], R["gen.text.never.logprobs"][:5])
    toks, lp = oq.generate_greedy(W, cfg, R["gen.image.input_ids"], torch.from_numpy(R["gen.image.pixel_values"]).to(BF),
                                  R["gen.image.grid_thw"], max_tokens=10, rope_mode="fallback", kv_bits=8, quantized_kv_start=0,
                                  return_logprobs=True)
    assert toks == R["gen.image.s0.tokens"].tolist()
    # (patch-embed Conv3d vs GEMM summation order: 1 bf16 ulp on a few image features, as in test_oracle_ref_golden.py; an
    #  8-bit code that flips on such an ulp moves a log-prob of magnitude 8..16 by a few of ITS ulps of 0.0625 - the text
    #  runs above are the bit-exact pin, this one checks the image path end to end)
    ref = R["gen.image.s0.logprobs"]
    d = lp.float().numpy() - ref
    assert np.abs(d).max() <= 0.5 and np.sqrt((d ** 2).mean() / (ref ** 2).mean()) < 5e-3, (np.abs(d).max(), np.sqrt((d ** 2).mean() / (ref ** 2).mean()))


def test_teacher_forced_decode_with_mid_stream_switch_matches_reference():
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    ids, forced = R["gen.text.input_ids"], R["tf.forced"]
    for tag, start in (("s0", 0), ("s26", 26)):
        got = oq.decode_teacher_forced(W, cfg, ids, forced_tokens=forced, rope_mode="fallback", kv_bits=8, kv_group_size=64,
                                       quantized_kv_start=start)
        assert np.array_equal(got.float().numpy(), R[f"tf.{tag}.logits"]), tag
    k = R["tf.s26.quantized_after_step"]
    assert k[:6].sum() == 0 and k[6:].all()          # 19 + 7 forwards = offset 26: every layer switches together


def test_w4_load_path_quantized_modules_and_logits_match_reference():
    """utils.py:736-987 on an MLX 4-bit checkpoint directory: the reference's predicate quantised exactly the modules whose
    `.scales` the checkpoint carries (the language model's 14 Linears + its embedding; the vision tower stays bf16), and the
    oracle's 4-bit forward (QW weights: QuantizedLinear, QuantizedEmbedding rows, tied head = as_linear) reproduces its logits."""
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    ck, ow = quant.quantize_checkpoint(W, lambda path, w: path.startswith("language_model."), 64, 4)
    mine = sorted(k[: -len(".weight")] for k, v in ow.items() if isinstance(v, quant.QW))
    assert mine == R["w4.quantized_paths"].tolist()
    ids, forced = R["gen.text.input_ids"], R["tf.forced"][:6]
    got, emb = oq.decode_teacher_forced(ow, cfg, ids, forced_tokens=forced, rope_mode="fallback", return_features=True)
    assert np.array_equal(emb[0].float().numpy(), R["w4.inputs_embeds"])
    assert np.array_equal(got.float().numpy(), R["w4.logits"])
    got = oq.decode_teacher_forced(ow, cfg, R["gen.image.input_ids"], torch.from_numpy(R["gen.image.pixel_values"]).to(BF),
                                   R["gen.image.grid_thw"], forced_tokens=(), rope_mode="fallback")
    ref = R["w4.image.prefill_last_logits"]          # (image path: the patch-embed summation-order ulp travels through 2 layers)
    d = got[0].float().numpy() - ref
    assert np.sqrt((d ** 2).mean() / (ref ** 2).mean()) < 1e-2
