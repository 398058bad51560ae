"""Golden vectors for the Phi-3.5-vision (`phi3_v`) path, produced by the REFERENCE'S OWN files (run once here).

    python tests/golden/make_golden_ref_phi3v.py      # needs /root/reference; writes tests/golden/phi3_v_tiny_ref.npz

Same method as make_golden_ref.py: `oracle/mlx_shim` stands in for `mlx`, and the reference's files

    mlx_vlm/models/phi3_v/{config,vision,phi3_v,processing_phi3_v}.py
    mlx_vlm/models/{base,cache,rope_utils,mlp,activations}.py, mlx_vlm/generate/ar.py (generate_step)

are imported unmodified from /root/reference and executed on the tiny config + seeded weights of oracle/phi3_v.py, in
fp32 and bf16.  The tower's width is a literal in the reference (CLIP ViT-L: 1024 / 16 heads / 336 px); only its depth and
MLP width are reduced, by replacing the `CLIP_VIT_LARGE_PATCH14_336_CONFIG` class attribute (a SimpleNamespace) before
the model is constructed - no reference file is edited.  Large activations are recorded on a subset of rows.  Only the
.npz travels to the GPU box.
"""
from __future__ import annotations

import importlib
import os
import sys
import types
import zlib
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ref as base  # noqa: E402  (path + shim bootstrap)

REF = base.REF
ROW_STRIDE = 11
CLIP_ROW_STRIDE, CLIP_COL_STRIDE = 37, 4          # the 1024-wide tower states: a 16 x 256 sample per view


def import_phi3v():
    mx, q, _cfgm, cache_mod, _su = base.import_reference()
    pkg = types.ModuleType("mlx_vlm.models.phi3_v")
    pkg.__path__ = [os.path.join(REF, "mlx_vlm", "models", "phi3_v")]
    sys.modules["mlx_vlm.models.phi3_v"] = pkg
    pm = importlib.import_module("mlx_vlm.models.phi3_v.phi3_v")
    cfgm = importlib.import_module("mlx_vlm.models.phi3_v.config")
    vis = importlib.import_module("mlx_vlm.models.phi3_v.vision")
    proc = importlib.import_module("mlx_vlm.models.phi3_v.processing_phi3_v")
    for m in (pm, cfgm, vis, proc):
        assert m.__file__.startswith(REF), m.__file__
    return mx, pm, cfgm, vis, proc, cache_mod, q._generate_ar


def ref_config(cfgm, cfg):
    t = cfg.text
    d = dict(model_type="phi3_v", vocab_size=t.vocab_size, num_hidden_layers=t.num_hidden_layers,
             intermediate_size=t.intermediate_size, num_attention_heads=t.num_attention_heads,
             num_key_value_heads=t.num_key_value_heads, rms_norm_eps=t.rms_norm_eps, hidden_size=t.hidden_size,
             rope_theta=t.rope_theta, max_position_embeddings=t.max_position_embeddings,
             original_max_position_embeddings=t.original_max_position_embeddings,
             rope_scaling={"type": "su", "short_factor": list(t.short_factor), "long_factor": list(t.long_factor)})
    mc = cfgm.ModelConfig.from_dict(d)
    if isinstance(mc.text_config, dict):
        mc.text_config = cfgm.TextConfig.from_dict(mc.text_config)
    if isinstance(mc.vision_config, dict):
        mc.vision_config = cfgm.VisionConfig.from_dict(mc.vision_config)
    return mc


class WordTokenizer:
    """deterministic stand-in for the HF tokenizer in the processor's prompt assembly: BOS + one id per whitespace word"""
    bos_token_id = 1
    pad_token_id = 0

    def encode(self, text, add_special_tokens=True):
        ids = [3 + (zlib.crc32(w.encode()) % 900) for w in text.split()]
        return ([self.bos_token_id] if add_special_tokens else []) + ids


def test_images():
    rng = np.random.default_rng(77)
    return [rng.integers(0, 256, (120, 200, 3), dtype=np.uint8),       # wide: 2 x 1 ... decided by the HD rule
            rng.integers(0, 256, (336, 336, 3), dtype=np.uint8),       # the benchmark's shape: 2 x 2 tiles at num_crops 4
            rng.integers(0, 256, (300, 90, 3), dtype=np.uint8)]        # tall


def main():
    from PIL import Image

    from oracle import phi3_v as op

    torch.manual_seed(0)
    torch.set_num_threads(8)
    mx, pm, cfgm, vis, proc_mod, cache_mod, ar = import_phi3v()
    f32 = base.f32
    cfg = op.tiny_cfg()
    mc = ref_config(cfgm, cfg)
    v = cfg.vision
    vis.VisionModel.CLIP_VIT_LARGE_PATCH14_336_CONFIG = SimpleNamespace(
        model_type="phi3_v", hidden_size=v.hidden_size, image_size=v.image_size, intermediate_size=v.intermediate_size,
        layer_norm_eps=v.layer_norm_eps, num_attention_heads=v.num_attention_heads, num_channels=v.num_channels,
        num_hidden_layers=v.num_hidden_layers, patch_size=v.patch_size)
    blob = {}

    # ---- the reference's image processor and prompt assembly (processing_phi3_v.py)
    imgs = test_images()
    ip = proc_mod.Phi3VImageProcessor()
    pix, sizes = [], []
    for i, im in enumerate(imgs):
        out = ip.preprocess([Image.fromarray(im)])
        pv = np.ascontiguousarray(np.asarray(out["pixel_values"]._t.numpy()))
        sz = np.asarray(out["image_sizes"]._t.numpy())
        pix.append(pv)
        sizes.append(sz)
        blob[f"img{i}.image_hwc"] = im
        blob[f"img{i}.ref_pixel_shape"] = np.array(pv.shape, dtype=np.int64)
        blob[f"img{i}.ref_pixel_dtype"] = np.array([str(pv.dtype)], dtype="U")
        blob[f"img{i}.ref_pixel_crc32"] = np.array([zlib.crc32(pv.astype(np.float32).tobytes())], dtype=np.int64)
        blob[f"img{i}.ref_pixel_sum"] = pv.astype(np.float64).sum(axis=(2, 3, 4))
        blob[f"img{i}.ref_image_sizes"] = sz.astype(np.int64)
        blob[f"img{i}.ref_num_tokens"] = np.array([ip.calc_num_image_tokens(Image.fromarray(im))], dtype=np.int64)
    both = ip.preprocess([Image.fromarray(imgs[0]), Image.fromarray(imgs[1])])
    blob["batch01.ref_pixel_shape"] = np.array(both["pixel_values"].shape, dtype=np.int64)
    blob["batch01.ref_image_sizes"] = np.asarray(both["image_sizes"]._t.numpy()).astype(np.int64)
    me = SimpleNamespace(image_processor=ip, tokenizer=WordTokenizer())
    text = "user says <|image_1|> what is this and <|image_2|> compare them please"
    res = proc_mod.Phi3VProcessor._convert_images_texts_to_inputs(me, [Image.fromarray(imgs[0]), Image.fromarray(imgs[1])], text)
    blob["prompt.text"] = np.array([text], dtype="U")
    blob["prompt.ref_input_ids"] = np.asarray(res["input_ids"]._t.numpy()).astype(np.int64)

    # ---- model cases
    rng = np.random.default_rng(5)

    def prompt(n_tokens_per_image):
        parts = [rng.integers(3, 1000, 5)]
        for j, n in enumerate(n_tokens_per_image):
            parts += [np.full(n, -(j + 1)), rng.integers(3, 1000, 4 + j)]
        return np.concatenate(parts).astype(np.int64)[None]

    cases = [([1], "one 336 x 336 image"), ([0, 2], "two images of different tile counts (padded views)")]
    case_ids = [prompt([int(ip.calc_num_image_tokens(Image.fromarray(imgs[i]))) for i in which]) for which, _ in cases]

    def bf16_bits(a):          # a bf16-valued array as uint16 (half the bytes of float32, exact)
        return (np.ascontiguousarray(f32(a)).view(np.uint32) >> 16).astype(np.uint16)
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        W = op.random_weights(cfg, seed=4321, dtype=torch.float32, **op.TEST_WEIGHT_SCALES)
        model = pm.Model(mc)
        weights = {k: mx.array(w.to(dt)) for k, w in W.items()}
        weights = model.vision_model.sanitize(weights)
        model.load_weights(list(weights.items()), strict=True)
        for ci, (which, _what) in enumerate(cases):
            p = f"case{ci}.{dt_name}."
            out = ip.preprocess([Image.fromarray(imgs[i]) for i in which])
            pv = np.asarray(out["pixel_values"]._t.numpy())
            sz = np.asarray(out["image_sizes"]._t.numpy())
            ids = case_ids[ci]
            input_ids = mx.array(ids.astype(np.int32))
            pixel_values = mx.array(pv.astype(np.float32))
            B, T = pv.shape[:2]
            tower_in = pixel_values.astype(dt).reshape(-1, *pv.shape[2:]).transpose(0, 2, 3, 1)
            _, _, states = model.vision_model.img_processor.vision_model(tower_in, True)
            emb = model.get_input_embeddings(input_ids, pixel_values, image_sizes=mx.array(sz)).inputs_embeds
            kv = [cache_mod.KVCache() for _ in model.layers]
            logits = model(input_ids, inputs_embeds=emb, cache=kv).logits

            def pick(lg):
                return mx.argmax(lg - mx.logsumexp(lg, axis=-1, keepdims=True), axis=-1)

            toks, step_logits = [], []
            y = pick(logits[:, -1, :])
            for _ in range(6):
                toks.append(int(y.item()))
                o = model(y[None] if y.ndim == 1 else y, cache=kv)
                step_logits.append(f32(o.logits[0, -1]))
                y = pick(o.logits[:, -1, :])
            if dt_name == "f32":
                blob[f"case{ci}.input_ids"] = ids
                blob[f"case{ci}.which"] = np.array(which, dtype=np.int64)
                blob[f"case{ci}.image_sizes"] = sz.astype(np.int64)
            if dt_name == "bf16":
                # complete where a bit-exact comparison needs complete inputs: view 0's pre_layrnorm output (a view's
                # encoder states depend on that view only) and the spliced prompt (the decoder is compared from there)
                if ci == 0:
                    blob[p + "ref_clip_embeddings_view0_bits"] = bf16_bits(states[0][0])
                blob[p + "ref_inputs_embeds_bits"] = bf16_bits(emb[0])
            blob[p + "ref_clip_state1"] = f32(states[1])[:, ::CLIP_ROW_STRIDE, ::CLIP_COL_STRIDE]
            blob[p + "ref_clip_feature_state"] = f32(states[-2])[:, ::CLIP_ROW_STRIDE, ::CLIP_COL_STRIDE]
            blob[p + "ref_inputs_embeds"] = f32(emb)[0, ::ROW_STRIDE]
            blob[p + "ref_inputs_embeds_len"] = np.array([emb.shape[1]], dtype=np.int64)
            blob[p + "ref_prefill_logits_last"] = f32(logits[0, -1])
            blob[p + "ref_decode_logits"] = np.stack(step_logits)
            blob[p + "ref_greedy"] = np.array(toks, dtype=np.int64)
            print(p, "emb", emb.shape, "greedy", toks)

    # ---- generate_step itself on the bf16 model: image prompt and text prompt
    ids = blob["case0.input_ids"]
    out = ip.preprocess([Image.fromarray(imgs[1])])
    toks, lps = [], []
    for tok, lp in ar.generate_step(mx.array(ids.astype(np.int32)), model, mx.array(np.asarray(out["pixel_values"]._t.numpy()).astype(np.float32)),
                                    None, max_tokens=6, temperature=0.0, image_sizes=out["image_sizes"]):
        toks.append(int(tok))
        lps.append(f32(lp))
    blob["generate_step.image.tokens"] = np.array(toks, dtype=np.int64)
    blob["generate_step.image.logprobs"] = np.stack(lps)
    text_ids = np.random.default_rng(12).integers(3, 1000, (1, 19)).astype(np.int32)
    toks, lps = [], []
    for tok, lp in ar.generate_step(mx.array(text_ids), model, None, None, max_tokens=6, temperature=0.0):
        toks.append(int(tok))
        lps.append(f32(lp))
    blob["generate_step.text.input_ids"] = text_ids.astype(np.int64)
    blob["generate_step.text.tokens"] = np.array(toks, dtype=np.int64)
    blob["generate_step.text.logprobs"] = np.stack(lps)
    print("generate_step image", blob["generate_step.image.tokens"].tolist(), "text", toks)

    # ---- SuScaledRoPE past original_max_position_embeddings (long factors) on the bf16 attention of layer 0
    x = mx.array(torch.randn(1, 2, 3, 96, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16))
    rope = model.layers[0].self_attn.rope
    blob["su_rope.x"] = f32(x)
    blob["su_rope.short_at_100"] = f32(rope(x, offset=100))
    blob["su_rope.long_at_4095"] = f32(rope(x, offset=4095))
    blob["su_rope.scale"] = np.array([float(rope._short_scale)], dtype=np.float64)
    # a BATCHED decode call, L = 1 (rope_utils.py:168-172: position_end = max(offset) + L decides for every row): row 0 at
    # offset 100 rides the long factors because row 1 sits at 4096 (4097 > 4096); a batch whose longest row is at 4095
    # (4096 > 4096 is false) stays short
    xb = mx.array(torch.randn(2, 2, 1, 96, generator=torch.Generator().manual_seed(4)).to(torch.bfloat16))
    blob["su_rope.xb"] = f32(xb)
    blob["su_rope.batched_100_4096"] = f32(rope(xb, offset=mx.array([100, 4096])))
    blob["su_rope.batched_100_4095"] = f32(rope(xb, offset=mx.array([100, 4095])))

    # ---- VisionModel.sanitize (vision.py:264-280): conv layout + position_ids
    hf = {"model.vision_embed_tokens.img_processor.vision_model.embeddings.patch_embedding.weight": mx.array(torch.zeros(8, 3, 14, 14)),
          "model.vision_embed_tokens.img_processor.vision_model.embeddings.position_ids": mx.array(torch.zeros(1, 4)),
          "model.embed_tokens.weight": mx.array(torch.zeros(4, 2)), "lm_head.weight": mx.array(torch.zeros(4, 2))}
    out = model.vision_model.sanitize(dict(hf))
    blob["sanitize.keys_in"] = np.array(sorted(hf), dtype="U")
    blob["sanitize.keys_out"] = np.array(sorted(out), dtype="U")
    blob["sanitize.conv_shape_out"] = np.array(
        out["model.vision_embed_tokens.img_processor.vision_model.embeddings.patch_embedding.weight"].shape, dtype=np.int64)
    blob["config.eos_token_id"] = np.array(mc.eos_token_id if mc.eos_token_id is not None else [], dtype=np.int64)

    path = os.path.join(HERE, "phi3_v_tiny_ref.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(blob), "arrays")


if __name__ == "__main__":
    main()
