"""Golden vectors for the Idefics2 path, produced by the REFERENCE'S OWN files (run once here).

    python tests/golden/make_golden_ref_idefics2.py      # needs /root/reference; writes tests/golden/idefics2_tiny_ref.npz

Same method as make_golden_ref.py: `oracle/mlx_shim` stands in for `mlx`, and the reference's files

    mlx_vlm/models/idefics2/{config,vision,language,idefics2}.py
    mlx_vlm/models/{base,cache,mlp,activations}.py, mlx_vlm/generate/ar.py (generate_step)

are imported unmodified from /root/reference and executed on the tiny config + seeded weights of oracle/idefics2.py, in
fp32 and bf16.  The HF processor wrapper (`processing_idefics2.py`) only registers a class at import: stubbed.  The image
processor the reference uses is transformers' own; its PIL backend is run here on the test images and recorded, so the
product's restatement is pinned without transformers' torchvision dependency at test time.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ref as base  # noqa: E402  (path + shim bootstrap)

REF = base.REF
ROW_STRIDE = 3


def import_idefics2():
    mx, q, _cfgm, cache_mod, _su = base.import_reference()
    pkg = types.ModuleType("mlx_vlm.models.idefics2")
    pkg.__path__ = [os.path.join(REF, "mlx_vlm", "models", "idefics2")]
    sys.modules["mlx_vlm.models.idefics2"] = pkg
    sys.modules["mlx_vlm.models.idefics2.processing_idefics2"] = types.ModuleType("mlx_vlm.models.idefics2.processing_idefics2")
    im = importlib.import_module("mlx_vlm.models.idefics2.idefics2")
    cfgm = importlib.import_module("mlx_vlm.models.idefics2.config")
    for m in (im, cfgm):
        assert m.__file__.startswith(REF), m.__file__
    return mx, im, cfgm, cache_mod, q._generate_ar


def ref_config(cfgm, cfg):
    t, v, p = cfg.text, cfg.vision, cfg.perceiver
    return cfgm.ModelConfig(
        text_config=cfgm.TextConfig(model_type="mistral", hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                                    num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                                    num_key_value_heads=t.num_key_value_heads, rms_norm_eps=t.rms_norm_eps,
                                    vocab_size=t.vocab_size, rope_theta=t.rope_theta),
        vision_config=cfgm.VisionConfig(model_type="idefics2", hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                                        num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                                        num_channels=v.num_channels, image_size=v.image_size, patch_size=v.patch_size,
                                        layer_norm_eps=v.layer_norm_eps),
        perceiver_config=cfgm.PerceiverConfig(model_type="idefics2", num_key_value_heads=p.num_key_value_heads,
                                              resampler_depth=p.resampler_depth, resampler_head_dim=p.resampler_head_dim,
                                              resampler_n_heads=p.resampler_n_heads, resampler_n_latents=p.resampler_n_latents),
        model_type="idefics2", image_token_id=cfg.image_token_id, vocab_size=t.vocab_size)


def test_images():
    rng = np.random.default_rng(91)
    return [rng.integers(0, 256, (90, 60, 3), dtype=np.uint8), rng.integers(0, 256, (56, 70, 3), dtype=np.uint8),
            rng.integers(0, 256, (336, 336, 3), dtype=np.uint8), rng.integers(0, 256, (300, 1200, 3), dtype=np.uint8)]


def main():
    from oracle import idefics2 as oi

    torch.manual_seed(0)
    torch.set_num_threads(8)
    mx, im, cfgm, cache_mod, ar = import_idefics2()
    f32 = base.f32
    cfg = oi.tiny_cfg()
    mc = ref_config(cfgm, cfg)
    imgs = test_images()
    blob = {f"img{i}.image_hwc": a for i, a in enumerate(imgs)}

    # ---- transformers' image processor (PIL backend), the one the reference's processor wraps
    from transformers.models.idefics2.image_processing_pil_idefics2 import Idefics2ImageProcessorPil

    for name, kw, which in (("default", {}, [[2], [3, 0]]), ("split", {"do_image_splitting": True}, [[2]]),
                            ("small", {"size": {"shortest_edge": 56, "longest_edge": 140}}, [[0, 1], [1]])):
        ip = Idefics2ImageProcessorPil(**kw)
        out = ip([[imgs[i] for i in row] for row in which], return_tensors="np")
        pv, pm = np.asarray(out["pixel_values"], dtype=np.float32), np.asarray(out["pixel_attention_mask"])
        blob[f"proc.{name}.which"] = np.array([r + [-1] * (2 - len(r)) for r in which], dtype=np.int64)
        blob[f"proc.{name}.pixel_shape"] = np.array(pv.shape, dtype=np.int64)
        blob[f"proc.{name}.pixel_crc32"] = np.array([zlib.crc32(np.ascontiguousarray(pv).tobytes())], dtype=np.int64)
        blob[f"proc.{name}.pixel_sum"] = pv.astype(np.float64).sum(axis=(2, 3, 4))
        blob[f"proc.{name}.mask_sum"] = pm.astype(np.int64).sum(axis=(2, 3))
        blob[f"proc.{name}.mask_crc32"] = np.array([zlib.crc32(np.ascontiguousarray(pm.astype(np.int64)).tobytes())], dtype=np.int64)

    # ---- model cases on the tiny tower (position table 10 x 10, patch 14): images preprocessed to <= 140 px
    rng = np.random.default_rng(6)
    nl = cfg.perceiver.resampler_n_latents

    def prompt(n_images):
        parts = [rng.integers(3, 1000, 5)]
        for j in range(n_images):
            parts += [np.full(nl, cfg.image_token_id), rng.integers(3, 1000, 3 + j)]
        return np.concatenate(parts).astype(np.int64)[None]

    cases = [[[0, 1]], [[1]]]                       # two images of different sizes in one prompt (padding + masks); one image
    case_inputs = []
    for which in cases:
        pv, pm = oi.preprocess([[imgs[i] for i in row] for row in which], shortest_edge=56, longest_edge=140)
        case_inputs.append((prompt(len(which[0])), pv, pm))

    def bf16_bits(a):
        return (np.ascontiguousarray(f32(a)).view(np.uint32) >> 16).astype(np.uint16)

    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        W = oi.random_weights(cfg, seed=4321, dtype=torch.float32, **oi.TEST_WEIGHT_SCALES)
        model = im.Model(mc)
        weights = {k: mx.array(w.to(dt)) for k, w in W.items()}
        weights = model.sanitize(weights)
        weights = model.language_model.sanitize(weights)
        weights = model.vision_model.sanitize(weights)
        model.load_weights(list(weights.items()), strict=True)
        for ci, (ids, pv, pm) in enumerate(case_inputs):
            p = f"case{ci}.{dt_name}."
            input_ids = mx.array(ids.astype(np.int32))
            pixel_values = mx.array(pv).astype(dt)
            mask = mx.array(pm)
            # the tower alone, called the way get_input_embeddings calls it
            B, N = pv.shape[:2]
            flat = pixel_values.reshape(B * N, *pv.shape[2:])
            pmask = oi.real_images_and_patch_mask(torch.from_numpy(pv), pm, cfg.vision.patch_size)[1]
            pooled, emb0, states = model.vision_model(flat.transpose(0, 2, 3, 1), patch_attention_mask=mx.array(pmask),
                                                      output_hidden_states=True)
            feats = model.connector(pooled.astype(pixel_values.dtype))
            emb = model.get_input_embeddings(input_ids, pixel_values, pixel_attention_mask=mask).inputs_embeds
            kv = [cache_mod.KVCache() for _ in model.language_model.layers]
            logits = model.language_model(input_ids, inputs_embeds=emb, cache=kv).logits

            def pick(lg):
                return mx.argmax(lg - mx.logsumexp(lg, axis=-1, keepdims=True), axis=-1)

            toks, step_logits = [], []
            y = pick(logits[:, -1, :])
            for _ in range(6):
                toks.append(int(y.item()))
                o = model.language_model(y[None] if y.ndim == 1 else y, cache=kv)
                step_logits.append(f32(o.logits[0, -1]))
                y = pick(o.logits[:, -1, :])
            if dt_name == "f32":
                blob[f"case{ci}.input_ids"] = ids
                blob[f"case{ci}.pixel_values"] = pv
                blob[f"case{ci}.pixel_attention_mask"] = pm.astype(np.int8)
                blob[f"case{ci}.patch_mask"] = pmask
            if dt_name == "bf16":
                blob[p + "ref_vision_embeddings_bits"] = bf16_bits(emb0)           # complete: the encoder is compared from here
                blob[p + "ref_inputs_embeds_bits"] = bf16_bits(emb[0])
            blob[p + "ref_vision_embeddings"] = f32(emb0)[:, ::ROW_STRIDE]
            blob[p + "ref_vision_layer0"] = f32(states[1])[:, ::ROW_STRIDE]
            blob[p + "ref_pooled"] = f32(pooled)[:, ::ROW_STRIDE]
            blob[p + "ref_image_features"] = f32(feats)
            blob[p + "ref_inputs_embeds"] = f32(emb)[0, ::ROW_STRIDE]
            blob[p + "ref_prefill_logits_last"] = f32(logits[0, -1])
            blob[p + "ref_decode_logits"] = np.stack(step_logits)
            blob[p + "ref_greedy"] = np.array(toks, dtype=np.int64)
            print(p, "pooled", pooled.shape, "feats", feats.shape, "emb", emb.shape, "greedy", toks)

    # ---- generate_step itself on the bf16 model: image prompt with bf16 pixels, the same with float32 pixels (as the
    # reference's own pipeline hands them over: the model never casts them), and a text prompt
    ids, pv, pm = case_inputs[0]
    for tag, pix in (("image", mx.array(pv).astype(torch.bfloat16)), ("image_f32_pixels", mx.array(pv))):
        toks, lps = [], []
        for tok, lp in ar.generate_step(mx.array(ids.astype(np.int32)), model, pix, None, max_tokens=6, temperature=0.0,
                                        pixel_attention_mask=mx.array(pm)):
            toks.append(int(tok))
            lps.append(f32(lp))
        blob[f"generate_step.{tag}.tokens"] = np.array(toks, dtype=np.int64)
        blob[f"generate_step.{tag}.logprobs"] = np.stack(lps)
        print("generate_step", tag, toks)
    text_ids = np.random.default_rng(12).integers(3, 1000, (1, 19)).astype(np.int32)
    toks, lps = [], []
    for tok, lp in ar.generate_step(mx.array(text_ids), model, None, None, max_tokens=6, temperature=0.0):
        toks.append(int(tok))
        lps.append(f32(lp))
    blob["generate_step.text.input_ids"] = text_ids.astype(np.int64)
    blob["generate_step.text.tokens"] = np.array(toks, dtype=np.int64)
    blob["generate_step.text.logprobs"] = np.stack(lps)
    print("generate_step text", toks)

    # ---- Model.sanitize (idefics2.py:294-321) + the conv layout (vision.py:207-222) on HF-layout key names
    hf = {"model.vision_model.embeddings.patch_embedding.weight": mx.array(torch.zeros(8, 3, 14, 14)),
          "model.vision_model.embeddings.patch_embedding.bias": mx.array(torch.zeros(8)),
          "model.connector.perceiver_resampler.latents": mx.array(torch.zeros(4, 2)),
          "model.connector.modality_projection.gate_proj.weight": mx.array(torch.zeros(2, 2)),
          "model.text_model.embed_tokens.weight": mx.array(torch.zeros(4, 2)),
          "model.text_model.layers.0.self_attn.q_proj.weight": mx.array(torch.zeros(2, 2)),
          "model.text_model.layers.0.self_attn.rotary_emb.inv_freq": mx.array(torch.zeros(2)),
          "model.text_model.norm.weight": mx.array(torch.zeros(2)), "lm_head.weight": mx.array(torch.zeros(4, 2))}
    out = model.vision_model.sanitize(model.language_model.sanitize(model.sanitize(dict(hf))))
    blob["sanitize.keys_in"] = np.array(sorted(hf), dtype="U")
    blob["sanitize.keys_out"] = np.array(sorted(out), dtype="U")
    blob["sanitize.conv_shape_out"] = np.array(out["vision_model.embeddings.patch_embedding.weight"].shape, dtype=np.int64)

    path = os.path.join(HERE, "idefics2_tiny_ref.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(blob), "arrays")


if __name__ == "__main__":
    main()
