"""Golden vectors for the QUANTISED paths, produced by the REFERENCE'S OWN Python files (run once, in the build container).

    python tests/golden/make_golden_ref_kvquant.py      # needs /root/reference; writes tests/golden/kvquant_ref.npz

Same method as make_golden_ref.py: `oracle/mlx_shim` stands in for `mlx`, the reference's files are imported UNMODIFIED
from /root/reference and executed.  New here: the shim's `mx.quantize / mx.dequantize / mx.quantized_matmul` and
`nn.quantize / nn.QuantizedLinear / nn.QuantizedEmbedding` (MLX's published affine algorithm, stated once in
oracle/quant.py - that arithmetic stays "parity unpinned"), which lets these reference files run:

    models/cache.py:233-334     QuantizedKVCache.update_and_fetch (256-step growth, slice assignment, quantise-on-update)
    models/cache.py:415-423     KVCache.to_quantized
    models/cache.py:8-21        should_quantize_kv_layer (the batch policy)
    models/base.py:260-302      quantized_scaled_dot_product_attention (typed `queries *= scale`, GQA expand, mask, softmax precise)
    models/base.py:305-373      scaled_dot_product_attention's dispatch on `hasattr(cache, "bits")`
    generate/common.py:77-181   maybe_quantize_kv_cache (uniform path: which layers switch, and when)
    generate/ar.py:151-515      generate_step(kv_bits=8, kv_group_size=64, quantized_kv_start=...)
    utils.py:736-987            load_model on an MLX 4-bit checkpoint directory: the nn.quantize class predicate
                                (utils.py:918-967), QuantizedLinear / QuantizedEmbedding forwards, tied head = as_linear

What is recorded: operator-level tensors (cache contents after two updates, to_quantized, the attention output for a
decode query and for a causal multi-row query), generate_step tokens + bf16 log-probs for three switch-over points, a
teacher-forced decode with the cache switching in the middle (every step's logits), and for the 4-bit checkpoint the set
of module paths the reference quantised plus prefill / decode logits.  tests/test_oracle_ref_golden_kvquant.py pins
oracle/quant.py and the oracle's 4-bit path to these bit for bit.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import make_golden_ref as G  # noqa: E402  (puts the shim on sys.path)

REF = G.REF
BF = torch.bfloat16


def u32(a):
    return np.asarray(a._t.contiguous().view(torch.int32).numpy()).view(np.uint32)


def tup(prefix, t, blob):
    blob[prefix + ".words"] = u32(t[0])
    blob[prefix + ".scales"] = G.f32(t[1])
    blob[prefix + ".biases"] = G.f32(t[2])


def build_model(mx, q, cfgm, cfg, W):
    mc = G.ref_config(cfgm, cfg)
    model = q.Model(mc)
    weights = {k: mx.array(w) for k, w in W.items()}
    weights = model.sanitize(weights) if hasattr(model, "sanitize") else weights
    weights = model.vision_tower.sanitize(weights)
    model.load_weights(list(weights.items()), strict=True)
    return model


def main():
    from oracle import quant as oquant
    from oracle import qwen2_vl as oq
    from make_golden import make_inputs

    torch.manual_seed(0)
    torch.set_num_threads(4)
    mx, q, cfgm, cache_mod, su = G.import_reference()
    import importlib

    base = importlib.import_module("mlx_vlm.models.base")
    common = importlib.import_module("mlx_vlm.generate.common")
    ar = q._generate_ar
    for m in (base, common, ar, cache_mod):
        assert m.__file__.startswith(REF), m.__file__
    blob = {}

    # ---------------------------------------------------------------- operators
    g = torch.Generator().manual_seed(41)
    Hq, Hkv, D = 4, 2, 128
    k1, v1 = (torch.randn(1, Hkv, 5, D, generator=g) * 1.5).to(BF), (torch.randn(1, Hkv, 5, D, generator=g) * 0.7).to(BF)
    k2, v2 = (torch.randn(1, Hkv, 1, D, generator=g) * 1.5).to(BF), (torch.randn(1, Hkv, 1, D, generator=g) * 0.7).to(BF)
    k1[0, 0, 2, :64] = 0.0                                            # a constant group: the 1e-7 scale floor, q0 == 0
    k1[0, 1, 3, 64:] = k1[0, 1, 3, 64:].abs()                         # an all-positive group (edge = w_max)
    qc = cache_mod.QuantizedKVCache(group_size=64, bits=8)
    s1 = qc.update_and_fetch(mx.array(k1), mx.array(v1))
    assert s1[0][0].shape[-2] == 5 and qc.keys[0].shape[-2] == 256    # the 256-step backing arrays, sliced to the offset
    s2 = qc.update_and_fetch(mx.array(k2), mx.array(v2))
    assert qc.offset == 6 and s2[0][0].shape[-2] == 6
    for n, t in (("k1", k1), ("v1", v1), ("k2", k2), ("v2", v2)):
        blob["op." + n] = t.float().numpy()
    tup("op.qcache.keys", s2[0], blob)
    tup("op.qcache.values", s2[1], blob)
    # KVCache.to_quantized after the same two updates (quantises the whole backing array; rows up to the offset matter)
    kc = cache_mod.KVCache()
    kc.update_and_fetch(mx.array(k1), mx.array(v1))
    kc.update_and_fetch(mx.array(k2), mx.array(v2))
    tq = kc.to_quantized(group_size=64, bits=8)
    assert tq.offset == 6
    st = tq.state if tq.keys[0].shape[2] == tq.offset else ([x[..., :6, :] for x in tq.keys], [x[..., :6, :] for x in tq.values])
    tup("op.to_quantized.keys", [x[..., :6, :] for x in tq.keys], blob)
    tup("op.to_quantized.values", [x[..., :6, :] for x in tq.values], blob)
    # quantized SDPA: one decode query over the 6 cached tokens; a 4-row causal query block over the same cache
    qd = (torch.randn(1, Hq, 1, D, generator=g)).to(BF)
    qm = (torch.randn(1, Hq, 4, D, generator=g)).to(BF)
    blob["op.q_decode"], blob["op.q_multi"] = qd.float().numpy(), qm.float().numpy()
    od = base.quantized_scaled_dot_product_attention(mx.array(qd.clone()), s2[0], s2[1], scale=D ** -0.5, mask=None, group_size=64, bits=8)
    om = base.quantized_scaled_dot_product_attention(mx.array(qm.clone()), s2[0], s2[1], scale=D ** -0.5, mask="causal", group_size=64, bits=8)
    blob["op.sdpa_decode"], blob["op.sdpa_causal"] = G.f32(od), G.f32(om)
    # ... and through the dispatcher (base.py:305-373): a cache object with `bits` routes to the quantized form
    od2 = base.scaled_dot_product_attention(mx.array(qd.clone()), s2[0], s2[1], qc, scale=D ** -0.5, mask=None)
    assert torch.equal(od2._t, od._t)
    blob["policy.should_quantize_kv_layer"] = np.array(
        [[n, i, int(cache_mod.should_quantize_kv_layer(i, n))] for n in (1, 2, 3, 28) for i in range(n)], dtype=np.int64)

    # ---------------------------------------------------------------- generate_step with kv_bits
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    model = build_model(mx, q, cfgm, cfg, W)
    text_ids = np.random.default_rng(21).integers(3, 1000, (1, 19)).astype(np.int32)
    blob["gen.text.input_ids"] = text_ids.astype(np.int64)
    imgs, pix, thw, ids = make_inputs(cfg, [(56, 84)], seed=1)
    blob["gen.image.input_ids"], blob["gen.image.pixel_values"], blob["gen.image.grid_thw"] = ids.astype(np.int64), pix.astype(np.float32), thw
    # the generate_step runs use the PEAKED head (tests/test_parity_decode_gpu.py's construction: an untied head whose next
    # token is a permutation successor with a wide margin) - with the seeded noise head greedy decoding sits on a fixed point
    cfgp = oq.tiny_cfg()
    cfgp.text.tie_word_embeddings = False
    Wp = oq.random_weights(cfgp, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    for k in list(Wp):
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            Wp[k] = (Wp[k].float() * 0.5).to(BF)
    Wp = oq.peak_head(Wp, cfgp, gamma=4.0, stride=389, n_cycle=1000)
    pmodel = build_model(mx, q, cfgm, cfgp, Wp)
    for tag, start in (("s0", 0), ("s24", 24), ("never", 10 ** 6)):
        toks, lps = [], []
        for tok, lp in ar.generate_step(mx.array(text_ids), pmodel, None, None, max_tokens=12, temperature=0.0, kv_bits=8,
                                        kv_group_size=64, quantized_kv_start=start):
            toks.append(int(tok))
            lps.append(G.f32(lp))
            assert lp.dtype == mx.bfloat16
        blob[f"gen.text.{tag}.tokens"], blob[f"gen.text.{tag}.logprobs"] = np.array(toks, dtype=np.int64), np.stack(lps)
        print("generate_step text", tag, toks)
    toks, lps = [], []
    for tok, lp in ar.generate_step(mx.array(ids.astype(np.int32)), pmodel, mx.array(pix.astype(np.float32)), None, max_tokens=10,
                                    temperature=0.0, image_grid_thw=mx.array(thw.astype(np.int32)), kv_bits=8, kv_group_size=64,
                                    quantized_kv_start=0):
        toks.append(int(tok))
        lps.append(G.f32(lp))
    blob["gen.image.s0.tokens"], blob["gen.image.s0.logprobs"] = np.array(toks, dtype=np.int64), np.stack(lps)
    print("generate_step image s0", toks)

    # ---------------------------------------------------------------- teacher-forced decode, the cache switching in the middle
    forced = np.random.default_rng(22).integers(3, 1000, 14)
    blob["tf.forced"] = forced.astype(np.int64)
    for tag, start in (("s0", 0), ("s26", 26)):
        kv = [cache_mod.KVCache() for _ in model.language_model.layers]
        emb = model.get_input_embeddings(mx.array(text_ids), None)
        out = model.language_model(mx.array(text_ids), inputs_embeds=emb.inputs_embeds, cache=kv)
        common.maybe_quantize_kv_cache(kv, start, 64, 8)
        rows, kinds = [G.f32(out.logits[0, -1])], []
        for y in forced:
            o = model.language_model(mx.array(np.array([[int(y)]], dtype=np.int32)), cache=kv)
            common.maybe_quantize_kv_cache(kv, start, 64, 8)
            rows.append(G.f32(o.logits[0, -1]))
            kinds.append([int(isinstance(c, cache_mod.QuantizedKVCache)) for c in kv])
        blob[f"tf.{tag}.logits"] = np.stack(rows)
        blob[f"tf.{tag}.quantized_after_step"] = np.array(kinds, dtype=np.int64)
        print("teacher-forced", tag, "first quantised after step", int(np.argmax(np.array(kinds)[:, 0])) if np.any(kinds) else None)

    # ---------------------------------------------------------------- the 4-bit load path (utils.py:736-987)
    # checkpoint directory in the layout mlx_vlm.convert leaves: config.json with "quantization", safetensors with
    # <path>.weight (uint32) / .scales / .biases for the language model's Linears and its embedding; vision tower in bf16
    from safetensors.torch import save_file

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m

    # utils.py's import-time dependencies that are not on the executed path
    G_pkg = types.ModuleType("mlx_vlm.quantization")
    G_pkg.__path__ = [os.path.join(REF, "mlx_vlm", "quantization")]
    sys.modules["mlx_vlm.quantization"] = G_pkg
    stub("mlx_vlm.quantization.one_bit", _quantization_for_path=lambda qz, p: qz, replace_one_bit_modules=lambda *a, **k: None)
    tr = types.ModuleType("mlx_vlm.trainer")
    tr.__path__ = []
    sys.modules["mlx_vlm.trainer"] = tr
    stub("mlx_vlm.trainer.utils", apply_lora_layers=None)
    del sys.modules["mlx_vlm.utils"]                                   # (make_golden_ref stubbed it for ar.py)
    from safetensors import safe_open

    def mx_load(path):
        out = {}
        with safe_open(path, framework="pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[k] = mx.array(t.view(torch.uint32) if t.dtype == torch.int32 and k.endswith(".weight") and (k[:-7] + ".scales") in f.keys() else t)
        return out

    mx.load = mx_load
    utils = importlib.import_module("mlx_vlm.utils")
    # make_golden_ref registered `mlx_vlm.models.qwen2_vl` as a bare package (its __init__ not executed); load_model resolves
    # the architecture through that package's namespace (utils.py:588-636), so run the package's own __init__ now
    pk = sys.modules["mlx_vlm.models.qwen2_vl"]
    init_py = os.path.join(REF, "mlx_vlm", "models", "qwen2_vl", "__init__.py")
    exec(compile(open(init_py).read(), init_py, "exec"), pk.__dict__)
    assert utils.__file__.startswith(REF)
    pred = lambda path, w: path.startswith("language_model.")          # noqa: E731  (what a text-only 4-bit conversion quantises)
    ck, ow = oquant.quantize_checkpoint(W, pred, 64, 4)
    t, v = cfg.text, cfg.vision
    config = dict(
        model_type="qwen2_vl", hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
        intermediate_size=t.intermediate_size, num_attention_heads=t.num_attention_heads,
        num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps,
        rope_theta=t.rope_theta, max_position_embeddings=32768, rope_scaling={"type": "mrope", "mrope_section": list(t.mrope_section)},
        tie_word_embeddings=t.tie_word_embeddings,
        vision_config=dict(model_type="qwen2_vl", depth=v.depth, embed_dim=v.embed_dim, hidden_size=v.hidden_size, num_heads=v.num_heads,
                           mlp_ratio=v.mlp_ratio, patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
                           temporal_patch_size=v.temporal_patch_size, in_channels=v.in_channels, skip_vision=True),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_token_id,
        quantization={"group_size": 64, "bits": 4},
    )
    with tempfile.TemporaryDirectory() as d:
        from pathlib import Path

        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(config, f)
        # the MLX-format checkpoint carries the conv weight already in MLX layout; sanitize() leaves such keys alone
        save_file({k: x.contiguous() for k, x in ck.items()}, os.path.join(d, "model.safetensors"))
        qmodel = utils.load_model(Path(d))
    import mlx.nn as nn

    qpaths = sorted(p for p, m in qmodel.named_modules() if isinstance(m, (nn.QuantizedLinear, nn.QuantizedEmbedding)))
    blob["w4.quantized_paths"] = np.array(qpaths)
    print("4-bit load: quantised modules", len(qpaths), qpaths[:3], "...")
    kv = [cache_mod.KVCache() for _ in qmodel.language_model.layers]
    emb = qmodel.get_input_embeddings(mx.array(text_ids), None)
    out = qmodel.language_model(mx.array(text_ids), inputs_embeds=emb.inputs_embeds, cache=kv)
    rows = [G.f32(out.logits[0, -1])]
    blob["w4.inputs_embeds"] = G.f32(emb.inputs_embeds[0])
    for y in forced[:6]:
        o = qmodel.language_model(mx.array(np.array([[int(y)]], dtype=np.int32)), cache=kv)
        rows.append(G.f32(o.logits[0, -1]))
    blob["w4.logits"] = np.stack(rows)
    # image features of the loaded model (the tower stays bf16) and the full path with an image
    emb = qmodel.get_input_embeddings(mx.array(ids.astype(np.int32)), mx.array(pix.astype(np.float32)), image_grid_thw=mx.array(thw.astype(np.int32)))
    kv = [cache_mod.KVCache() for _ in qmodel.language_model.layers]
    out = qmodel.language_model(mx.array(ids.astype(np.int32)), inputs_embeds=emb.inputs_embeds, cache=kv, position_ids=emb.position_ids,
                                rope_deltas=emb.rope_deltas)
    blob["w4.image.prefill_last_logits"] = G.f32(out.logits[0, -1])

    out_path = os.path.join(HERE, "kvquant_ref.npz")
    np.savez_compressed(out_path, **blob)
    print("wrote", out_path, os.path.getsize(out_path), "bytes;", len(blob), "arrays")


if __name__ == "__main__":
    main()
