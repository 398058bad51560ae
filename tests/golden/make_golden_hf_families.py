"""A second, independent pin for the three families whose only judge so far was the reference run over `oracle/mlx_shim`
(VERDICT r04, item 5): HuggingFace transformers (5.15.0) on torch-CPU fp32, as `make_golden.py` does for Qwen2-VL.

    python tests/golden/make_golden_hf_families.py        # writes tests/golden/families_hf.npz

What HF has a model class for is what is pinned:
  idefics2   `Idefics2ForConditionalGeneration` - the whole path: tower (with the patch attention mask of two images of
             different sizes), perceiver resampler + modality projection, Mistral decoder -> image features and logits.
  nanoLLaVA  `SiglipVisionModel` (the tower, hidden state of the layer the reference selects) and `Qwen2ForCausalLM`
             run on the oracle's merged input embeddings -> features and logits.  The 2-layer GELU projector and the merge
             are the only parts with no HF class behind them (they stay pinned by the shim goldens only).
  phi3_v     `CLIPVisionModel` (hidden state -2, CLS dropped) and `Phi3ForCausalLM` with the `longrope` scaling, run on the
             oracle's merged input embeddings.  The HD transform (sub-image layout, separators) has no HF class here.
mlx-vlm loads the very same checkpoints as HF (renamed keys, transposed conv weights), so HF fp32 pins the model
semantics; it does not pin MLX's bf16 rounding (that is the shim goldens' job).  Weights are not stored: they are
regenerated from the oracle's seeded `random_weights` and loaded into the HF modules by name.
"""
from __future__ import annotations

import os
import sys
import warnings
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
F32 = torch.float32


def _load(m, sd, allow_missing=()):
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    bad = [k for k in missing if not any(a in k for a in ("rotary", "inv_freq", "position_ids") + tuple(allow_missing))]
    assert not bad, bad
    return m.eval().to(F32)


# ------------------------------------------------------------------------------------------------------ idefics2
def idefics2(blob):
    from transformers import Idefics2Config, Idefics2ForConditionalGeneration

    from oracle import idefics2 as oi

    cfg = oi.tiny_cfg()
    t, v, p = cfg.text, cfg.vision, cfg.perceiver
    hc = Idefics2Config(
        text_config=dict(model_type="mistral", hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                         num_hidden_layers=t.num_hidden_layers, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, rms_norm_eps=t.rms_norm_eps, vocab_size=t.vocab_size,
                         rope_theta=t.rope_theta, sliding_window=None, head_dim=t.head_dim, max_position_embeddings=4096,
                         pad_token_id=0),
        vision_config=dict(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                           num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads, num_channels=3,
                           image_size=v.image_size, patch_size=v.patch_size, layer_norm_eps=v.layer_norm_eps,
                           hidden_act="quick_gelu"),
        perceiver_config=dict(resampler_n_latents=p.resampler_n_latents, resampler_depth=p.resampler_depth,
                              resampler_n_heads=p.resampler_n_heads, resampler_head_dim=p.resampler_head_dim,
                              num_key_value_heads=p.num_key_value_heads, hidden_act="silu", hidden_size=t.hidden_size,
                              rms_norm_eps=t.rms_norm_eps),
        image_token_id=cfg.image_token_id, tie_word_embeddings=False, pad_token_id=0)
    hc._attn_implementation = "eager"
    W = oi.random_weights(cfg, seed=4321, dtype=F32, **oi.TEST_WEIGHT_SCALES)
    sd = {}
    for k, w in W.items():
        if k.startswith("language_model.lm_head."):
            sd["lm_head." + k[len("language_model.lm_head."):]] = w
        elif k.startswith("language_model."):
            sd["model.text_model." + k[len("language_model."):]] = w
        else:
            if k.endswith("patch_embedding.weight"):
                w = w.permute(0, 3, 1, 2).contiguous()              # (O, kH, kW, C) -> (O, C, kH, kW)
            sd["model." + k] = w
    m = _load(Idefics2ForConditionalGeneration(hc), sd)

    # The reference departs from HF's Idefics2 in three places; it is the spec, so the HF modules are driven its way and
    # each departure is a line here, not a tolerance:
    #   1. vision MLP activation: x * sigmoid(1.702 x) (`FastGELUMLP`, idefics2/vision.py:7) where the checkpoint's config
    #      says gelu_pytorch_tanh  -> hidden_act="quick_gelu" above;
    #   2. position ids: np.digitize(..., right=True) - 1 (idefics2/vision.py:150-166) gives -1 for the first row / column
    #      (a negative index: the LAST row of the table) and other bucket edges than HF's bucketize -> the ids come from
    #      the oracle's restatement of those lines; the conv and the table lookup are HF's modules;
    #   3. the encoder and the resampler get no attention mask (padding patches of the smaller image are attended:
    #      idefics2/vision.py:186-205, idefics2.py:163-177) and post_layernorm uses nn.LayerNorm's default eps 1e-5.
    vm = m.model.vision_model
    vm.post_layernorm.eps = 1e-5
    rng = np.random.default_rng(91)
    imgs = [rng.integers(0, 256, (90, 60, 3), dtype=np.uint8), rng.integers(0, 256, (56, 70, 3), dtype=np.uint8)]
    nl = p.resampler_n_latents
    prng = np.random.default_rng(6)
    for name, which in (("two", [0, 1]), ("one", [1])):
        pv, pm = oi.preprocess([[imgs[i] for i in which]], shortest_edge=56, longest_edge=140)
        parts = [prng.integers(3, 1000, 5)]
        for j in range(len(which)):
            parts += [np.full(nl, cfg.image_token_id), prng.integers(3, 1000, 3 + j)]
        ids = torch.from_numpy(np.concatenate(parts).astype(np.int64)[None])
        pvt, pmask = oi.real_images_and_patch_mask(torch.as_tensor(np.asarray(pv), dtype=F32), pm, v.patch_size)
        pos = torch.from_numpy(oi.position_ids(pmask, v.num_patches_per_side)) % (v.num_patches_per_side ** 2)
        with torch.no_grad():
            emb = vm.embeddings.patch_embedding(pvt).flatten(2).transpose(1, 2) + vm.embeddings.position_embedding(pos)
            hs = vm.post_layernorm(vm.encoder(inputs_embeds=emb).last_hidden_state)
            feats = m.model.connector(hs, attention_mask=torch.ones(hs.shape[:2], dtype=torch.bool))
            merged = m.model.inputs_merger(ids, m.model.text_model.embed_tokens(ids), feats)
            logits = m.lm_head(m.model.text_model(inputs_embeds=merged, use_cache=False).last_hidden_state)
            greedy = [int(logits[0, -1].argmax())]
        blob[f"idefics2.{name}.input_ids"] = ids.numpy()
        blob[f"idefics2.{name}.which"] = np.array(which, dtype=np.int64)
        blob[f"idefics2.{name}.hf_tower"] = hs.numpy()
        blob[f"idefics2.{name}.hf_image_features"] = feats.numpy()
        blob[f"idefics2.{name}.hf_logits"] = logits[0].numpy()


def _greedy_with_cache(lm, embed, inputs_embeds, n):
    """HF decoder on given prompt embeddings, then n - 1 one-token steps through HF's own KV cache.  -> prompt logits
    [L, V], greedy tokens, the logits row behind each token"""
    with torch.no_grad():
        out = lm(inputs_embeds=inputs_embeds, use_cache=True)
        prompt_logits = out.logits[0]
        rows, toks = [out.logits[0, -1]], []
        past = out.past_key_values
        for i in range(n):
            toks.append(int(rows[-1].argmax()))
            if i == n - 1:
                break
            out = lm(inputs_embeds=embed(torch.tensor([[toks[-1]]])), past_key_values=past, use_cache=True)
            past = out.past_key_values
            rows.append(out.logits[0, -1])
    return prompt_logits.numpy(), np.array(toks, dtype=np.int64), torch.stack(rows).numpy()


# ------------------------------------------------------------------------------------------------------ nanoLLaVA
def nanollava(blob):
    from transformers import Qwen2Config, Qwen2ForCausalLM, SiglipVisionConfig, SiglipVisionModel

    from oracle import llava_bunny as ob

    cfg = ob.tiny_cfg()
    t, v = cfg.text, cfg.vision
    W = ob.random_weights(cfg, seed=1234, dtype=F32, **ob.TEST_WEIGHT_SCALES)
    # tower.  Reference departure 1 again: FastGELUMLP (llava_bunny/vision.py over mlp.py:47-57) = quick_gelu where the
    # SigLIP checkpoint's config says gelu_pytorch_tanh.  The model takes the LAST ENCODER STATE, before post_layernorm
    # (`*_, hidden_state = vision_tower(..., output_hidden_states=True)`; `hidden_state[-1]`, llava_bunny.py:113-117).
    vc = SiglipVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                            num_attention_heads=v.num_attention_heads, num_channels=3, image_size=v.image_size,
                            patch_size=v.patch_size, layer_norm_eps=v.layer_norm_eps, hidden_act="quick_gelu")
    vc._attn_implementation = "eager"
    sd = {}
    for k, w in W.items():
        if k.startswith(ob.V):
            if k.endswith("patch_embedding.weight"):
                w = w.permute(0, 3, 1, 2).contiguous()
            sd[k[len(ob.V):]] = w
    tower = _load(SiglipVisionModel(vc), sd, allow_missing=("head.",))
    qc = Qwen2Config(hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers, intermediate_size=t.intermediate_size,
                     num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                     rms_norm_eps=t.rms_norm_eps, vocab_size=t.vocab_size, rope_theta=t.rope_theta,
                     tie_word_embeddings=t.tie_word_embeddings, max_position_embeddings=4096, use_sliding_window=False,
                     bos_token_id=0, eos_token_id=1, pad_token_id=2)
    qc._attn_implementation = "eager"
    sd = {"model." + k[len(ob.LM):]: w for k, w in W.items() if k.startswith(ob.LM)}
    if t.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    lm = _load(Qwen2ForCausalLM(qc), sd)

    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    pix = torch.from_numpy(ob.preprocess([img]))
    ids = np.concatenate([rng.integers(3, 1000, 6), [cfg.image_token_index], rng.integers(3, 1000, 5)]).astype(np.int64)[None]
    with torch.no_grad():
        hs = tower(pixel_values=pix, output_hidden_states=True).hidden_states[-1]
    blob["nanollava.image_crc32"] = np.array([zlib.crc32(img.tobytes())], dtype=np.int64)    # rng(17)'s first draw, (300, 420, 3)
    blob["nanollava.input_ids"] = ids
    blob["nanollava.hf_tower_last_state"] = hs.numpy()
    # projector + splice: no HF class; the oracle's (fp32) on HF's tower output, so the decoder pin starts from HF features
    feats = ob.mm_projector(W, hs)
    emb = ob.embed_tokens(W, ids)
    pos = int(np.argmax(ids[0] == cfg.image_token_index))
    merged = torch.cat([emb[:, :pos], feats, emb[:, pos + 1:]], dim=1)
    pl, toks, rows = _greedy_with_cache(lm, lm.model.embed_tokens, merged, 6)
    blob["nanollava.hf_prompt_logits_last8"] = pl[-8:]
    blob["nanollava.hf_greedy"] = toks
    blob["nanollava.hf_greedy_logits"] = rows


# ------------------------------------------------------------------------------------------------------ phi3_v
def phi3v(blob):
    from transformers import CLIPVisionConfig, CLIPVisionModel, Phi3Config, Phi3ForCausalLM

    from oracle import phi3_v as op

    base = op.tiny_cfg()
    v = base.vision
    W = op.random_weights(base, seed=2024, dtype=F32, **op.TEST_WEIGHT_SCALES)
    # tower: CLIP ViT-L/14-336's own activation is quick_gelu, so FastGELUMLP is no departure here.  The HD transform reads
    # hidden state -2 with the class row dropped (phi3_v/vision.py:224-226).
    cc = CLIPVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size, num_hidden_layers=v.num_hidden_layers,
                          num_attention_heads=v.num_attention_heads, num_channels=3, image_size=v.image_size,
                          patch_size=v.patch_size, layer_norm_eps=v.layer_norm_eps, hidden_act="quick_gelu", projection_dim=64)
    cc._attn_implementation = "eager"
    sd = {}
    for k, w in W.items():
        if k.startswith(op.CLIP):
            if k.endswith("patch_embedding.weight"):
                w = w.permute(0, 3, 1, 2).contiguous()
            sd[k[len(op.CLIP):]] = w
    tower = _load(CLIPVisionModel(cc), sd)
    rng = np.random.default_rng(23)
    img = rng.integers(0, 256, (200, 500, 3), dtype=np.uint8)
    pix, sizes = op.preprocess([img])[:2]
    pix_t = torch.as_tensor(np.asarray(pix), dtype=F32)
    crops = pix_t.reshape(-1, *pix_t.shape[-3:])
    with torch.no_grad():
        hs = tower(pixel_values=crops, output_hidden_states=True).hidden_states[-2][:, 1:]
    blob["phi3v.image_crc32"] = np.array([zlib.crc32(img.tobytes())], dtype=np.int64)        # rng(23)'s first draw, (200, 500, 3)
    blob["phi3v.hf_clip_state_m2_sub"] = hs[:, ::6, ::4].numpy().copy()      # every 6th patch, every 4th channel (295 KB)

    # decoder: Phi3ForCausalLM with the longrope scaling, in both regimes - `short` (original_max 4096: every position of
    # the test is below it) and `long` (original_max 16 < prompt length: the long factors and the same attention factor).
    # HF scales cos / sin by sqrt(1 + ln(factor) / ln(original_max)); the reference multiplies q and k by it before the
    # rotation (rope_utils.py:168-189) - the same number on both sides of the dot product.
    for name, omax, pmax in (("short", 4096, 131072), ("long", 16, 64)):
        cfg = op.tiny_cfg()
        t = cfg.text
        t.original_max_position_embeddings, t.max_position_embeddings = omax, pmax
        pc = Phi3Config(hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers, intermediate_size=t.intermediate_size,
                        num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                        rms_norm_eps=t.rms_norm_eps, vocab_size=t.vocab_size, max_position_embeddings=pmax,
                        original_max_position_embeddings=omax, tie_word_embeddings=False, sliding_window=None,
                        rope_parameters=dict(rope_type="longrope", rope_theta=t.rope_theta, short_factor=t.short_factor,
                                             long_factor=t.long_factor, original_max_position_embeddings=omax,
                                             factor=pmax / omax),
                        bos_token_id=0, eos_token_id=1, pad_token_id=2)
        pc._attn_implementation = "eager"
        sd = {k: w for k, w in W.items() if not k.startswith(op.VT)}
        lm = _load(Phi3ForCausalLM(pc), sd)
        n_img = op.num_image_tokens(img.shape[1], img.shape[0])
        ids = np.concatenate([rng.integers(3, 1000, 4), np.full(n_img, -1), rng.integers(3, 1000, 5)]).astype(np.int64)[None]
        # image features -> prompt embeddings: the HD transform and the splice have no HF class (oracle's, in fp32, on the
        # tower states HF just produced)
        merged = op.get_input_embeddings(W, cfg, ids, pix_t, sizes)
        pl, toks, rows = _greedy_with_cache(lm, lm.model.embed_tokens, merged, 6)
        blob[f"phi3v.{name}.input_ids"] = ids
        blob[f"phi3v.{name}.rope"] = np.array([omax, pmax], dtype=np.int64)
        blob[f"phi3v.{name}.hf_prompt_logits_last8"] = pl[-8:]
        blob[f"phi3v.{name}.hf_greedy"] = toks
        blob[f"phi3v.{name}.hf_greedy_logits"] = rows


FAMILIES = []


def main():
    warnings.filterwarnings("ignore")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    blob = {}
    for fam in FAMILIES:
        fam(blob)
    np.savez_compressed(os.path.join(OUT, "families_hf.npz"), **blob)
    print("wrote families_hf.npz:", {k: v.shape for k, v in blob.items()})



FAMILIES[:] = [idefics2, nanollava, phi3v]

if __name__ == "__main__":
    main()
