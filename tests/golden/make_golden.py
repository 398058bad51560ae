"""Generate the golden fixtures under tests/golden/ (run once, in the build
container; the fixtures are committed, this script is how they were made).

    python tests/golden/make_golden.py

Source of truth: HuggingFace transformers (5.15.0) `Qwen2VLForConditionalGeneration`
on torch-CPU fp32.  mlx-vlm loads the very same checkpoints (it only renames
keys and transposes the conv weight: qwen2_vl.py:179-190, vision.py:292-310), so
HF fp32 outputs pin the *model semantics* of the path; they do not pin MLX's
bf16 rounding ("parity unpinned", oracle/__init__.py).  The reference itself
cannot run here (no `mlx`).

Weights are NOT stored: they are regenerated from a seed by
`oracle.qwen2_vl.random_weights` (torch CPU generator, deterministic) and
loaded into the HF model through the inverse of `sanitize`.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import image_processor as ip  # noqa: E402
from oracle import qwen2_vl as oq  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def hf_model_from_oracle_weights(cfg: oq.Cfg, W):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration

    t, v = cfg.text, cfg.vision
    hcfg = Qwen2VLConfig(
        text_config=dict(hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
                         intermediate_size=t.intermediate_size, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size,
                         rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta,
                         rope_scaling={"type": "mrope", "mrope_section": t.mrope_section},
                         tie_word_embeddings=t.tie_word_embeddings, max_position_embeddings=32768,
                         bos_token_id=0, eos_token_id=1, pad_token_id=2),
        vision_config=dict(depth=v.depth, embed_dim=v.embed_dim, hidden_size=v.hidden_size, num_heads=v.num_heads,
                           mlp_ratio=int(v.mlp_ratio), patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
                           temporal_patch_size=v.temporal_patch_size, in_channels=v.in_channels),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
        vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_start_token_id + 1,
        tie_word_embeddings=t.tie_word_embeddings, bos_token_id=0, eos_token_id=1, pad_token_id=2,
    )
    hcfg._attn_implementation = "eager"
    m = Qwen2VLForConditionalGeneration(hcfg).eval().to(torch.float32)
    sd = {}
    for k, w in W.items():
        w = w.to(torch.float32)
        if k.startswith("vision_tower."):
            hk = "model.visual." + k[len("vision_tower."):]
            if "patch_embed.proj.weight" in k:
                w = w.permute(0, 4, 1, 2, 3).contiguous()  # (O,T,H,W,C) -> (O,C,T,H,W)
        elif k.startswith("language_model.model."):
            hk = "model.language_model." + k[len("language_model.model."):]
        elif k.startswith("language_model.lm_head."):
            hk = "lm_head." + k[len("language_model.lm_head."):]
        else:
            raise KeyError(k)
        sd[hk] = w
    if t.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.language_model.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in k or "inv_freq" in k for k in missing), missing
    return m


def make_inputs(cfg: oq.Cfg, sizes, n_text=12, seed=0):
    rng = np.random.default_rng(seed)
    imgs = [rng.integers(0, 256, (3, h, w), dtype=np.uint8) for (h, w) in sizes]
    pix, thw = ip.process(imgs)
    ids = []
    for _ in imgs:
        ids += [cfg.vision_start_token_id, cfg.image_token_id, cfg.vision_start_token_id + 1]
        ids += rng.integers(3, 1000, 3).tolist()
    ids += rng.integers(3, 1000, n_text).tolist()
    ids = ip.expand_image_placeholders(ids, cfg.image_token_id, thw)
    return imgs, pix, thw, np.array([ids], dtype=np.int64)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=torch.float32, std=0.05, embed_std=0.2)
    hf = hf_model_from_oracle_weights(cfg, W)

    cases = {"one_image": [(56, 84)], "two_images": [(56, 56), (84, 56)]}
    blob = {}
    for name, sizes in cases.items():
        imgs, pix, thw, ids = make_inputs(cfg, sizes, seed=len(sizes))
        with torch.no_grad():
            tid = torch.from_numpy(ids)
            tpix = torch.from_numpy(pix)
            tthw = torch.from_numpy(thw)
            mm = (tid == cfg.image_token_id).to(torch.int32)  # image == 1 (HF 5.x processor output)
            out = hf(input_ids=tid, pixel_values=tpix, image_grid_thw=tthw, mm_token_type_ids=mm, use_cache=False)
            logits = out.logits[0].numpy()
            feats = hf.model.visual(tpix, grid_thw=tthw)
            if not torch.is_tensor(feats):
                feats = feats.pooler_output if hasattr(feats, "pooler_output") else feats[0]
            feats = feats.numpy()
            pos, deltas = hf.model.get_rope_index(tid, mm_token_type_ids=mm, image_grid_thw=tthw)
            gen = hf.generate(input_ids=tid, pixel_values=tpix, image_grid_thw=tthw, mm_token_type_ids=mm,
                              max_new_tokens=8,
                              do_sample=False, eos_token_id=None, pad_token_id=2)
        blob[name + ".images"] = np.concatenate([im.reshape(-1) for im in imgs])
        blob[name + ".sizes"] = np.array(sizes, dtype=np.int64)
        blob[name + ".pixel_values"] = pix.astype(np.float32)
        blob[name + ".grid_thw"] = thw
        blob[name + ".input_ids"] = ids
        blob[name + ".hf_logits"] = logits.astype(np.float32)
        blob[name + ".hf_image_features"] = feats.astype(np.float32)
        blob[name + ".hf_position_ids"] = pos.numpy().astype(np.int64)
        blob[name + ".hf_rope_deltas"] = deltas.numpy().astype(np.int64)
        blob[name + ".hf_greedy"] = gen[0, ids.shape[1]:].numpy().astype(np.int64)
        print(name, ids.shape, pix.shape, thw.tolist(), logits.shape, blob[name + ".hf_greedy"])

    # text-only + padded batch rope-index tables (integer goldens from HF)
    tid = torch.tensor([[5, 6, 7, 8, 9, 10], [2, 2, 11, 12, 13, 14]])
    am = torch.tensor([[1, 1, 1, 1, 1, 1], [0, 0, 1, 1, 1, 1]])
    try:
        pos, deltas = hf.model.get_rope_index(tid, mm_token_type_ids=torch.zeros_like(tid, dtype=torch.int32),
                                              attention_mask=am)
        blob["text_padded.input_ids"] = tid.numpy()
        blob["text_padded.attention_mask"] = am.numpy()
        blob["text_padded.hf_position_ids"] = pos.numpy().astype(np.int64)
        blob["text_padded.hf_rope_deltas"] = deltas.numpy().astype(np.int64)
    except Exception as e:  # HF 5.x moved text-only handling elsewhere
        print("text-only rope index not available from HF:", type(e).__name__, e)

    # HF image processor goldens (integer/f32 exact): smart_resize + patchify
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import smart_resize as hf_smart_resize

    sr = []
    for (h, w) in [(336, 336), (448, 448), (100, 333), (1080, 1920), (30, 40), (2000, 3000), (57, 500)]:
        sr.append([h, w, *hf_smart_resize(h, w, factor=28, min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280)])
    blob["smart_resize.table"] = np.array(sr, dtype=np.int64)

    # HF PIL image processor (Qwen2VLImageProcessorPil): pixel_values / grid for two images
    try:
        from PIL import Image
        from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil

        proc = Qwen2VLImageProcessorPil(image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5],
                                        min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280)
        rng = np.random.default_rng(7)
        for tag, (h, w) in {"ip_a": (100, 150), "ip_b": (336, 336)}.items():
            im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            o = proc(images=[Image.fromarray(im)], return_tensors="np")
            blob[tag + ".image_hwc"] = im
            pv = np.asarray(o["pixel_values"], dtype=np.float32)
            # keep the fixture small: full values only for the small image, per-row sums for all
            blob[tag + ".hf_pixel_values_rowsum"] = pv.astype(np.float64).sum(axis=1)
            blob[tag + ".hf_pixel_values"] = pv if pv.shape[0] <= 128 else pv[:0]
            blob[tag + ".hf_grid_thw"] = np.asarray(o["image_grid_thw"], dtype=np.int64)
            print(tag, blob[tag + ".hf_pixel_values"].shape, blob[tag + ".hf_grid_thw"].tolist())
    except Exception as e:
        print("HF PIL image processor golden skipped:", type(e).__name__, e)

    np.savez_compressed(os.path.join(OUT, "qwen2_vl_tiny_hf.npz"), **blob)
    print("wrote", os.path.join(OUT, "qwen2_vl_tiny_hf.npz"),
          os.path.getsize(os.path.join(OUT, "qwen2_vl_tiny_hf.npz")), "bytes")


if __name__ == "__main__":
    main()
