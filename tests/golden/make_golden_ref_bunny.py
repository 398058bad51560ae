"""Golden vectors for the nanoLLaVA (`llava_bunny`) path, produced by the REFERENCE'S OWN files (run once here).

    python tests/golden/make_golden_ref_bunny.py      # needs /root/reference; writes tests/golden/llava_bunny_tiny_ref.npz

Same method as make_golden_ref.py: `oracle/mlx_shim` stands in for `mlx`, and the reference's files

    mlx_vlm/models/llava_bunny/{config,vision,language,llava_bunny}.py   (incl. its ImageProcessor, llava_bunny.py:24-57)
    mlx_vlm/models/{base,cache,mlp,activations}.py, mlx_vlm/generate/ar.py (generate_step)

are imported unmodified from /root/reference and executed on the tiny config + seeded weights of
oracle/llava_bunny.py, in fp32 and bf16.  Large activations are recorded on a fixed subset of rows (every 7th of the
729 patch rows) to keep the fixture small; logits and tokens are complete.  Only the .npz travels to the GPU box.
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
ROW_STRIDE = 7


def import_bunny():
    mx, q, _cfgm, cache_mod, _su = base.import_reference()
    pkg = types.ModuleType("mlx_vlm.models.llava_bunny")
    pkg.__path__ = [os.path.join(REF, "mlx_vlm", "models", "llava_bunny")]
    sys.modules["mlx_vlm.models.llava_bunny"] = pkg
    # `from ..llava import processing_llava` only registers an HF processor class: not on the executed path
    llava = types.ModuleType("mlx_vlm.models.llava")
    llava.processing_llava = types.ModuleType("mlx_vlm.models.llava.processing_llava")
    sys.modules["mlx_vlm.models.llava"] = llava
    sys.modules["mlx_vlm.models.llava.processing_llava"] = llava.processing_llava
    lb = importlib.import_module("mlx_vlm.models.llava_bunny.llava_bunny")
    cfgm = importlib.import_module("mlx_vlm.models.llava_bunny.config")
    for m in (lb, cfgm):
        assert m.__file__.startswith(REF), m.__file__
    return mx, lb, cfgm, cache_mod, q._generate_ar


def ref_config(cfgm, cfg):
    t, v = cfg.text, cfg.vision
    d = dict(model_type="llava_bunny", auto_map={}, hidden_size=t.hidden_size, mm_hidden_size=v.hidden_size,
             num_hidden_layers=t.num_hidden_layers, intermediate_size=t.intermediate_size,
             num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
             rms_norm_eps=t.rms_norm_eps, vocab_size=t.vocab_size, rope_theta=t.rope_theta,
             attention_bias=t.attention_bias, tie_word_embeddings=t.tie_word_embeddings,
             image_token_index=cfg.image_token_index,
             vision_config=dict(num_hidden_layers=v.num_hidden_layers, hidden_size=v.hidden_size,
                                intermediate_size=v.intermediate_size, num_attention_heads=v.num_attention_heads,
                                image_size=v.image_size, patch_size=v.patch_size, num_channels=v.num_channels,
                                layer_norm_eps=v.layer_norm_eps))
    mc = cfgm.ModelConfig.from_dict(d)
    if isinstance(mc.text_config, dict):
        mc.text_config = cfgm.TextConfig.from_dict(mc.text_config)
    if isinstance(mc.vision_config, dict):
        mc.vision_config = cfgm.VisionConfig.from_dict(mc.vision_config)
    return mc


def test_inputs(cfg):
    """Two seeded images (not 384 x 384: the processor resizes) and prompts with the <image> sentinel inside."""
    rng = np.random.default_rng(2024)
    imgs = [rng.integers(0, 256, (100, 150, 3), dtype=np.uint8), rng.integers(0, 256, (200, 150, 3), dtype=np.uint8)]
    ids = [np.concatenate([rng.integers(3, 1000, 5), [cfg.image_token_index], rng.integers(3, 1000, 7)]),
           np.concatenate([[cfg.image_token_index], rng.integers(3, 1000, 9)])]
    return imgs, [i.astype(np.int64)[None] for i in ids]


def main():
    from PIL import Image

    from oracle import llava_bunny as ob

    torch.manual_seed(0)
    torch.set_num_threads(4)
    mx, lb, cfgm, cache_mod, ar = import_bunny()
    f32 = base.f32
    cfg = ob.tiny_cfg()
    mc = ref_config(cfgm, cfg)
    imgs, ids_list = test_inputs(cfg)
    blob = {}

    # ---- the reference's ImageProcessor (llava_bunny.py:24-57 over base.py:121-194)
    proc = lb.ImageProcessor()
    pix = []
    for i, im in enumerate(imgs):
        out = proc.preprocess([Image.fromarray(im)])
        pv = np.ascontiguousarray(np.asarray(out[0], dtype=np.float32))
        assert pv.shape == (3, 384, 384), pv.shape
        pix.append(pv)
        blob[f"img{i}.image_hwc"] = im
        blob[f"img{i}.ref_pixel_crc32"] = np.array([zlib.crc32(pv.tobytes())], dtype=np.int64)
        blob[f"img{i}.ref_pixel_rowsum"] = pv.astype(np.float64).sum(axis=(0, 2))

    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        W = ob.random_weights(cfg, seed=4321, dtype=torch.float32, with_pooling_head=True, **ob.TEST_WEIGHT_SCALES)
        model = lb.Model(mc)
        weights = {k: mx.array(w.to(dt)) for k, w in W.items()}
        weights = model.sanitize(weights)
        weights = model.language_model.sanitize(weights)
        weights = model.vision_tower.vision_tower.sanitize(weights)
        model.load_weights(list(weights.items()), strict=True)
        for i, (pv, ids) in enumerate(zip(pix, ids_list)):
            p = f"case{i}.{dt_name}."
            input_ids = mx.array(ids.astype(np.int32))
            pixel_values = mx.array(pv[None]).astype(dt)
            _, last, states = model.vision_tower(pixel_values.transpose(0, 2, 3, 1), output_hidden_states=True)
            feats = model.mm_projector(states[-1].astype(pixel_values.dtype))
            emb = model.get_input_embeddings(input_ids, pixel_values).inputs_embeds
            kv = [cache_mod.KVCache() for _ in model.language_model.layers]
            logits = model.language_model(input_ids, inputs_embeds=emb, cache=kv).logits
            def pick(lg):      # generate_step's greedy rule (ar.py:368-379): argmax of logits - logsumexp, in the model dtype
                return mx.argmax(lg - mx.logsumexp(lg, axis=-1, keepdims=True), axis=-1)

            toks, step_logits = [], []
            y = pick(logits[:, -1, :])
            for _ in range(6):
                toks.append(int(y.item()))
                o = model.language_model(y[None] if y.ndim == 1 else y, cache=kv)
                step_logits.append(f32(o.logits[0, -1]))
                y = pick(o.logits[:, -1, :])
            blob[p + "ref_embeddings"] = f32(states[0])[0, ::ROW_STRIDE]
            if dt_name == "bf16":      # complete: the encoder comparison starts from these (see the test)
                blob[p + "ref_embeddings_full"] = f32(states[0])[0]
            blob[p + "ref_layer0"] = f32(states[1])[0, ::ROW_STRIDE]
            blob[p + "ref_vision_last"] = f32(states[-1])[0, ::ROW_STRIDE]
            blob[p + "ref_image_features"] = f32(feats)[0, ::ROW_STRIDE]
            blob[p + "ref_inputs_embeds"] = f32(emb)[0, ::ROW_STRIDE]
            blob[p + "ref_inputs_embeds_len"] = np.array([emb.shape[1]], dtype=np.int64)
            blob[p + "ref_prefill_logits_last"] = f32(logits[0, -1])
            blob[p + "ref_prefill_logits_rows"] = f32(logits[0])[::97]
            blob[p + "ref_decode_logits"] = np.stack(step_logits)
            blob[p + "ref_greedy"] = np.array(toks, dtype=np.int64)
            blob[p + "ref_kv_offset"] = np.array([kv[0].offset], dtype=np.int64)
            if dt_name == "f32":
                blob[f"case{i}.input_ids"] = ids
            print(p, "emb", emb.shape, "greedy", toks)

    # ---- generate_step itself (generate/ar.py:151-515), greedy, on the bf16 model; image prompt and text prompt
    toks, lps = [], []
    for tok, lp in ar.generate_step(mx.array(ids_list[0].astype(np.int32)), model, mx.array(pix[0][None]), None,
                                    max_tokens=6, temperature=0.0):
        toks.append(int(tok))
        lps.append(f32(lp))
    blob["generate_step.image.tokens"] = np.array(toks, dtype=np.int64)
    blob["generate_step.image.logprobs"] = np.stack(lps)
    text_ids = np.random.default_rng(12).integers(3, 1000, (1, 17)).astype(np.int32)
    toks, lps = [], []
    for tok, lp in ar.generate_step(mx.array(text_ids), model, None, None, max_tokens=6, temperature=0.0):
        toks.append(int(tok))
        lps.append(f32(lp))
    blob["generate_step.text.input_ids"] = text_ids.astype(np.int64)
    blob["generate_step.text.tokens"] = np.array(toks, dtype=np.int64)
    blob["generate_step.text.logprobs"] = np.stack(lps)
    print("generate_step image", blob["generate_step.image.tokens"].tolist(), "text", toks)

    # ---- checkpoint key remap (Model.sanitize llava_bunny.py:180-222; vision.py:243-266 conv layout)
    hf = {"model.vision_tower.vision_tower.vision_model.embeddings.patch_embedding.weight": mx.array(torch.zeros(8, 3, 14, 14)),
          "model.vision_tower.vision_tower.vision_model.embeddings.position_ids": mx.array(torch.zeros(1, 4)),
          "model.vision_tower.vision_tower.vision_model.head.attention.in_proj_weight": mx.array(torch.zeros(6, 2)),
          "model.vision_tower.vision_tower.vision_model.head.attention.in_proj_bias": mx.array(torch.zeros(6)),
          "model.mm_projector.0.weight": mx.array(torch.zeros(2, 2)), "model.mm_projector.0.bias": mx.array(torch.zeros(2)),
          "model.mm_projector.2.weight": mx.array(torch.zeros(2, 2)), "model.mm_projector.2.bias": mx.array(torch.zeros(2)),
          "model.embed_tokens.weight": mx.array(torch.zeros(4, 2)), "model.norm.weight": mx.array(torch.zeros(2)),
          "model.layers.0.self_attn.q_proj.weight": mx.array(torch.zeros(2, 2)),
          "model.layers.0.self_attn.rotary_emb.inv_freq": mx.array(torch.zeros(2)),
          "lm_head.weight": mx.array(torch.zeros(4, 2))}
    out = model.vision_tower.vision_tower.sanitize(model.language_model.sanitize(model.sanitize(dict(hf))))
    blob["sanitize.keys_in"] = np.array(sorted(hf), dtype="U")
    blob["sanitize.keys_out"] = np.array(sorted(out), dtype="U")
    blob["sanitize.conv_shape_out"] = np.array(
        out["vision_tower.vision_tower.vision_model.embeddings.patch_embedding.weight"].shape, dtype=np.int64)

    path = os.path.join(HERE, "llava_bunny_tiny_ref.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(blob), "arrays")


if __name__ == "__main__":
    main()
