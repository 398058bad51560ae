"""Golden vectors produced by the REFERENCE'S OWN Python files (run once, in the build container).

    python tests/golden/make_golden_ref.py        # needs /root/reference; writes tests/golden/qwen2_vl_tiny_ref.npz

The reference (mlx-vlm) is pure Python over Apple's `mlx`, which cannot be installed here.  This script puts
`oracle/mlx_shim` (a torch-CPU stand-in for mlx.core / mlx.nn / mlx.utils, see its README) on sys.path and then
imports and RUNS the reference's files for the Qwen2-VL path unchanged from /root/reference:

    mlx_vlm/models/qwen2_vl/{config,vision,language,qwen2_vl}.py
    mlx_vlm/models/{base,cache,rope_utils,mlp,activations}.py
    mlx_vlm/sample_utils.py, mlx_vlm/generate/ar.py (generate_step) + generate/common.py

on the tiny config + seeded weights of tests/golden/make_golden.py, in fp32 and in bf16.  The package
`__init__`s of mlx_vlm are NOT executed (they pull the whole server / tokenizer stack); `mlx_vlm.turboquant`
and the HF-processor patch module are stubbed - neither is on the executed path.

What is recorded per case: vision-tower output, merged input embeddings, position ids + rope deltas, prefill
logits, 8 greedy decode steps through the reference's KVCache (tokens + per-step logits); plus rope-index tables
for text-only / left-padded batches and the sampler filters (top-k / top-p / min-p) on fixed log-probs.
Nothing here is read at test time on the GPU box: only the .npz travels.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VLM_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "mlx_shim"))


def import_reference():
    """-> (mx, qwen2_vl module, cache module, sample_utils module) from the reference tree, unmodified."""
    import mlx.core as mx  # the shim

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__package__ = name
        sys.modules[name] = m
        return m

    pkg("mlx_vlm", os.path.join(REF, "mlx_vlm"))
    pkg("mlx_vlm.models", os.path.join(REF, "mlx_vlm", "models"))
    pkg("mlx_vlm.models.qwen2_vl", os.path.join(REF, "mlx_vlm", "models", "qwen2_vl"))
    # not on the executed path: KV-quantisation codecs and the HF processor patch
    tq = types.ModuleType("mlx_vlm.turboquant")
    for n in ("BatchTurboQuantKVCache", "TurboQuantKVCache", "HybridQuantKVCache"):
        setattr(tq, n, type(n, (), {}))
    tq._state_length = lambda s: s[0].shape[-2]
    tq.turboquant_enabled = lambda *a, **k: False
    sys.modules["mlx_vlm.turboquant"] = tq
    sys.modules["mlx_vlm.models.qwen2_vl.processing_qwen2_vl"] = types.ModuleType(
        "mlx_vlm.models.qwen2_vl.processing_qwen2_vl")
    # generate/ar.py (generate_step) drags in the server / tokenizer / speculative stack at import time; none of it
    # is executed by generate_step for a plain greedy run, so those modules are empty stand-ins
    pkg("mlx_vlm.generate", os.path.join(REF, "mlx_vlm", "generate"))
    pkg("mlx_vlm.speculative", os.path.join(REF, "mlx_vlm", "_not_imported_"))

    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m

    stub("mlx_vlm.apc")
    stub("mlx_vlm.kv_quant", from_legacy=lambda *a, **k: None)
    stub("mlx_vlm.prompt_utils", apply_chat_template=None)
    stub("mlx_vlm.speculative.utils", make_speculative_prompt_cache=None, run_speculative_rounds=None,
         run_speculative_server_rounds=None, speculative_hidden_state=None,
         speculative_prefill_kwargs=lambda *a, **k: {})
    stub("mlx_vlm.utils", group_images_by_shape=None, prepare_inputs=None, should_add_special_tokens=None)
    import importlib

    q = importlib.import_module("mlx_vlm.models.qwen2_vl.qwen2_vl")
    cfgm = importlib.import_module("mlx_vlm.models.qwen2_vl.config")
    cache = importlib.import_module("mlx_vlm.models.cache")
    su = importlib.import_module("mlx_vlm.sample_utils")
    ar = importlib.import_module("mlx_vlm.generate.ar")
    for m in (q, cfgm, cache, su, ar):
        assert m.__file__.startswith(REF), m.__file__
    q._generate_ar = ar
    return mx, q, cfgm, cache, su


def ref_config(cfgm, cfg):
    t, v = cfg.text, cfg.vision
    # HF Qwen2-VL config.json layout: text parameters at the root (config.py:71-77 copies them into text_config)
    d = dict(
        model_type="qwen2_vl", hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
        intermediate_size=t.intermediate_size, num_attention_heads=t.num_attention_heads,
        num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps,
        rope_theta=t.rope_theta, max_position_embeddings=32768,
        rope_scaling={"type": "mrope", "mrope_section": list(t.mrope_section)},
        tie_word_embeddings=t.tie_word_embeddings,
        vision_config=dict(model_type="qwen2_vl", depth=v.depth, embed_dim=v.embed_dim, hidden_size=v.hidden_size,
                           num_heads=v.num_heads, mlp_ratio=v.mlp_ratio, patch_size=v.patch_size,
                           spatial_merge_size=v.spatial_merge_size, temporal_patch_size=v.temporal_patch_size,
                           in_channels=v.in_channels),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
        vision_start_token_id=cfg.vision_start_token_id,
    )
    mc = cfgm.ModelConfig.from_dict(d)
    if isinstance(mc.text_config, dict):
        mc.text_config = cfgm.TextConfig.from_dict(mc.text_config)
    if isinstance(mc.vision_config, dict):
        mc.vision_config = cfgm.VisionConfig.from_dict(mc.vision_config)
    return mc


def f32(a):
    return np.asarray(a._t.to(torch.float32).numpy()) if hasattr(a, "_t") else np.asarray(a, dtype=np.float32)


def main():
    from oracle import qwen2_vl as oq
    sys.path.insert(0, HERE)
    from make_golden import make_inputs

    torch.manual_seed(0)
    torch.set_num_threads(4)
    mx, q, cfgm, cache_mod, su = import_reference()
    cfg = oq.tiny_cfg()
    mc = ref_config(cfgm, cfg)
    blob = {}
    cases = {"one_image": [(56, 84)], "two_images": [(56, 56), (84, 56)]}
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        W = oq.random_weights(cfg, seed=1234, dtype=torch.float32, std=0.05, embed_std=0.2)
        model = q.Model(mc)
        weights = {k: mx.array(w.to(dt)) for k, w in W.items()}
        weights = model.sanitize(weights) if hasattr(model, "sanitize") else weights
        weights = model.vision_tower.sanitize(weights)
        model.load_weights(list(weights.items()), strict=True)
        for name, sizes in cases.items():
            imgs, pix, thw, ids = make_inputs(cfg, sizes, seed=len(sizes))
            input_ids = mx.array(ids.astype(np.int32))
            pixel_values = mx.array(pix.astype(np.float32))
            grid = mx.array(thw.astype(np.int32))
            feats = model.vision_tower(pixel_values.astype(dt), grid, output_hidden_states=False)
            patches = model.vision_tower.patch_embed(pixel_values.astype(dt))
            emb = model.get_input_embeddings(input_ids, pixel_values, image_grid_thw=grid)
            kv = [cache_mod.KVCache() for _ in model.language_model.layers]
            out = model.language_model(input_ids, inputs_embeds=emb.inputs_embeds, cache=kv, position_ids=emb.position_ids,
                                       pixel_values=pixel_values, image_grid_thw=grid, rope_deltas=emb.rope_deltas)
            # the reference's generate_step hands position_ids only for the prompt; decode positions come from
            # cache offset + rope_deltas (language.py:476-509).  Drive that loop exactly as ar.py:_step does.
            logits = out.logits
            step_logits, toks = [], []
            y = mx.argmax(logits[:, -1, :], axis=-1)
            for _ in range(8):
                toks.append(int(y.item()))
                o = model.language_model(y[None] if y.ndim == 1 else y, cache=kv)
                step_logits.append(f32(o.logits[0, -1]))
                y = mx.argmax(o.logits[:, -1, :], axis=-1)
            p = f"{name}.{dt_name}."
            blob[p + "ref_image_features"] = f32(feats)
            blob[p + "ref_patch_embed"] = f32(patches)
            blob[p + "ref_inputs_embeds"] = f32(emb.inputs_embeds[0])
            blob[p + "ref_prefill_logits"] = f32(logits[0])
            blob[p + "ref_decode_logits"] = np.stack(step_logits)
            blob[p + "ref_greedy"] = np.array(toks, dtype=np.int64)
            blob[p + "ref_kv_offset"] = np.array([kv[0].offset], dtype=np.int64)
            if dt_name == "f32":
                blob[name + ".ref_position_ids"] = np.asarray(emb.position_ids._t.numpy()).astype(np.int64)
                blob[name + ".ref_rope_deltas"] = np.asarray(emb.rope_deltas._t.numpy()).astype(np.int64).reshape(-1)
                blob[name + ".input_ids"] = ids
                blob[name + ".grid_thw"] = thw
                blob[name + ".sizes"] = np.array(sizes, dtype=np.int64)
            print(p, "feats", blob[p + "ref_image_features"].shape, "logits", blob[p + "ref_prefill_logits"].shape,
                  "greedy", toks)

    # ---- the reference's generate_step itself (generate/ar.py:151-515), greedy, bf16 model (the last one built)
    ar = q._generate_ar
    imgs, pix, thw, ids = make_inputs(cfg, cases["one_image"], seed=1)
    toks, lps = [], []
    for tok, lp in ar.generate_step(mx.array(ids.astype(np.int32)), model, mx.array(pix.astype(np.float32)), None,
                                    max_tokens=8, temperature=0.0, image_grid_thw=mx.array(thw.astype(np.int32))):
        toks.append(int(tok))
        lps.append(f32(lp))
        assert lp.dtype == mx.bfloat16
    blob["generate_step.image.tokens"] = np.array(toks, dtype=np.int64)
    blob["generate_step.image.logprobs"] = np.stack(lps)
    text_ids = np.random.default_rng(11).integers(3, 1000, (1, 19)).astype(np.int32)
    toks, lps = [], []
    for tok, lp in ar.generate_step(mx.array(text_ids), model, None, None, max_tokens=8, temperature=0.0):
        toks.append(int(tok))
        lps.append(f32(lp))
    blob["generate_step.text.input_ids"] = text_ids.astype(np.int64)
    blob["generate_step.text.tokens"] = np.array(toks, dtype=np.int64)
    blob["generate_step.text.logprobs"] = np.stack(lps)
    print("generate_step image", blob["generate_step.image.tokens"].tolist(), "text", toks)

    # ---- rope index: text only, and a left-padded batch (language.py:216-402)
    lm = model.language_model
    tid = mx.array(np.array([[5, 6, 7, 8, 9, 10], [2, 2, 11, 12, 13, 14]], dtype=np.int32))
    am = mx.array(np.array([[1, 1, 1, 1, 1, 1], [0, 0, 1, 1, 1, 1]], dtype=np.int32))
    pos, delta = lm.get_rope_index(tid, attention_mask=am)
    blob["text_padded.input_ids"] = np.asarray(tid._t.numpy()).astype(np.int64)
    blob["text_padded.attention_mask"] = np.asarray(am._t.numpy()).astype(np.int64)
    blob["text_padded.ref_position_ids"] = np.asarray(pos._t.numpy()).astype(np.int64)
    blob["text_padded.ref_rope_deltas"] = np.asarray(delta._t.numpy()).astype(np.int64).reshape(-1)
    pos, delta = lm.get_rope_index(tid)
    blob["text_only.ref_position_ids"] = np.asarray(pos._t.numpy()).astype(np.int64)
    blob["text_only.ref_rope_deltas"] = np.asarray(delta._t.numpy()).astype(np.int64).reshape(-1)
    # image + left padding in one batch row (mask-aware path, language.py:236-374)
    imgs, pix, thw, ids = make_inputs(cfg, [(56, 56)], seed=5)
    row = ids[0].tolist()
    padded = np.array([[2, 2, 2] + row], dtype=np.int32)
    mask = np.array([[0, 0, 0] + [1] * len(row)], dtype=np.int32)
    pos, delta = lm.get_rope_index(mx.array(padded), mx.array(thw.astype(np.int32)), None, mx.array(mask))
    blob["image_padded.input_ids"] = padded.astype(np.int64)
    blob["image_padded.attention_mask"] = mask.astype(np.int64)
    blob["image_padded.grid_thw"] = thw
    blob["image_padded.ref_position_ids"] = np.asarray(pos._t.numpy()).astype(np.int64)
    blob["image_padded.ref_rope_deltas"] = np.asarray(delta._t.numpy()).astype(np.int64).reshape(-1)

    # ---- sampler filters on fixed log-probs (sample_utils.py:149-175,266-318)
    g = torch.Generator().manual_seed(77)
    logits = torch.randn(3, 257, generator=g) * 3.0
    lp = logits - torch.logsumexp(logits, -1, keepdim=True)
    blob["sampler.logprobs"] = lp.numpy()
    blob["sampler.top_k_5"] = f32(su.apply_top_k(mx.array(lp), 5))
    blob["sampler.top_p_0.9"] = f32(su.apply_top_p(mx.array(lp), 0.9))
    blob["sampler.top_p_0.5"] = f32(su.apply_top_p(mx.array(lp), 0.5))
    blob["sampler.min_p_0.05"] = f32(su.apply_min_p(mx.array(lp), 0.05))
    blob["sampler.greedy"] = np.asarray(su.make_sampler(temp=0.0)(mx.array(lp))._t.numpy()).astype(np.int64)

    # ---- the reference's own image processor (models/qwen3_vl/processing_qwen3_vl.py:182-205,302-378; Qwen2-VL
    #      uses it with patch 14): smart resize table + patchified pixel rows for the two HF-golden test images
    import importlib
    import zlib

    m = types.ModuleType("mlx_vlm.models.qwen3_vl")
    m.__path__ = [os.path.join(REF, "mlx_vlm", "models", "qwen3_vl")]
    sys.modules["mlx_vlm.models.qwen3_vl"] = m
    ip3 = importlib.import_module("mlx_vlm.models.qwen3_vl.processing_qwen3_vl")
    assert ip3.__file__.startswith(REF)
    from PIL import Image

    G = np.load(os.path.join(HERE, "qwen2_vl_tiny_hf.npz"))
    proc = ip3.Qwen3VLImageProcessor(patch_size=14, merge_size=2, temporal_patch_size=2, min_pixels=56 * 56,
                                     max_pixels=14 * 14 * 4 * 1280, image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5])
    for tag in ("ip_a", "ip_b"):
        o = proc([Image.fromarray(G[tag + ".image_hwc"])])
        pv = np.ascontiguousarray(np.asarray(o["pixel_values"], dtype=np.float32))
        blob[tag + ".ref_grid_thw"] = np.asarray(o["image_grid_thw"], dtype=np.int64)
        blob[tag + ".ref_pixel_values"] = pv if pv.shape[0] <= 128 else pv[:0]
        blob[tag + ".ref_pixel_values_rowsum"] = pv.astype(np.float64).sum(axis=1)
        blob[tag + ".ref_pixel_values_crc32"] = np.array([zlib.crc32(pv.tobytes())], dtype=np.int64)
    sr = [[h, w, *ip3._smart_resize_image(h, w, factor=28, min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280)]
          for (h, w) in [(336, 336), (448, 448), (100, 333), (1080, 1920), (30, 40), (2000, 3000), (57, 500)]]
    blob["smart_resize.ref_table"] = np.array(sr, dtype=np.int64)

    # ---- checkpoint key remap + conv-weight layout (Model.sanitize qwen2_vl.py:179-190, VisionModel.sanitize
    #      vision.py:292-310) on the HF Qwen2-VL checkpoint key layout
    hf_keys = (["visual.patch_embed.proj.weight", "visual.merger.ln_q.weight", "visual.merger.mlp.0.bias",
                "visual.blocks.0.attn.qkv.weight", "visual.blocks.1.mlp.fc2.bias", "visual.blocks.0.norm1.weight",
                "model.embed_tokens.weight", "model.norm.weight", "lm_head.weight",
                "model.layers.0.self_attn.q_proj.bias", "model.layers.1.mlp.down_proj.weight",
                "model.layers.0.input_layernorm.weight", "model.layers.1.post_attention_layernorm.weight"])
    gen = torch.Generator().manual_seed(5)
    hf_w = {k: mx.array(torch.randn(4, 3, 2, 6, 6, generator=gen) if k.endswith("patch_embed.proj.weight")
                        else torch.randn(3, generator=gen)) for k in hf_keys}
    san = model.vision_tower.sanitize(model.sanitize(dict(hf_w)))
    blob["sanitize.hf_keys"] = np.array(hf_keys)
    blob["sanitize.ref_keys"] = np.array(list(san.keys()))
    blob["sanitize.hf_conv"] = f32(hf_w["visual.patch_embed.proj.weight"])
    blob["sanitize.ref_conv"] = f32(san["vision_tower.patch_embed.proj.weight"])

    # ---- NaiveStreamingDetokenizer (tokenizer_utils.py:71-118; needs no mlx) on a byte-level toy tokenizer: multi-byte
    #      UTF-8 characters arrive split across tokens, newlines flush
    tu = importlib.import_module("mlx_vlm.tokenizer_utils")
    assert tu.__file__.startswith(REF)

    class ByteTok:
        def decode(self, toks):
            return bytes(toks).decode("utf-8", errors="replace")

    text = "héllo wörld\n日本語 ok\nfin 🙂!"
    toks = list(text.encode("utf-8"))
    det = tu.NaiveStreamingDetokenizer(ByteTok())
    det.reset()
    segs = []
    for t in toks:
        det.add_token(t)
        segs.append(det.last_segment)
    det.finalize()
    segs.append(det.last_segment)
    blob["detok.tokens"] = np.array(toks, dtype=np.int64)
    blob["detok.ref_segments"] = np.array(segs)
    blob["detok.ref_text"] = np.array([det.text])

    out = os.path.join(HERE, "qwen2_vl_tiny_ref.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes;", len(blob), "arrays")


if __name__ == "__main__":
    main()
