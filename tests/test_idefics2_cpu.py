"""Idefics2 host logic on the CPU (no kernel is launched): image processor and prompt expansion against the oracle / the
reference's rule, model construction and weight packing without a device, padding-image removal and patch masks,
load_processor / prepare_inputs plumbing."""
import json

import numpy as np
import pytest
import torch

from oracle import idefics2 as oi

BF = torch.bfloat16


def test_image_processor_bit_exact_vs_oracle():
    """the product's table-driven processor == the oracle's restatement (itself crc-exact vs transformers' PIL backend)"""
    from mlx_vlm_amd.models.idefics2 import Idefics2ImageProcessor

    rng = np.random.default_rng(2)
    imgs = [rng.integers(0, 256, s, dtype=np.uint8) for s in ((336, 336, 3), (200, 420, 3), (90, 60, 3), (300, 1300, 3))]
    for kw, okw in (({}, {}), ({"do_image_splitting": True}, {"do_image_splitting": True}),
                    ({"size": {"shortest_edge": 56, "longest_edge": 140}}, {"shortest_edge": 56, "longest_edge": 140})):
        ip = Idefics2ImageProcessor(**kw)
        for sample in ([[imgs[0]]], [[imgs[1], imgs[2]], [imgs[3]]], [[imgs[0], imgs[0], imgs[0], imgs[0]]]):
            got = ip(sample)
            pv, pm = oi.preprocess(sample, **okw)
            assert got["pixel_values"].dtype == np.float32 and np.array_equal(got["pixel_values"], pv)
            assert np.array_equal(got["pixel_attention_mask"], pm)
    # the benchmark's shape: 336 x 336 is raised to 378 x 378 -> 27 x 27 patches
    assert Idefics2ImageProcessor()([[imgs[0]]])["pixel_values"].shape == (1, 1, 3, 378, 378)


def test_prompt_expansion_follows_the_reference_rule():
    """processing_idefics2.py:105-127: <image> -> fake + 64 x <image> + fake (x 5 when splitting), doubled fake tokens merged,
    a space after a closing fake token that is followed by text"""
    from mlx_vlm_amd.models.idefics2 import Idefics2ImageProcessor, Idefics2Processor

    class Tok:
        def convert_tokens_to_ids(self, t):
            return 7

        def __call__(self, texts, **kw):
            return {"input_ids": [[len(t)] for t in texts], "attention_mask": [[1] for _ in texts]}

    F, I = "<fake_token_around_image>", "<image>"
    p = Idefics2Processor(Idefics2ImageProcessor(), Tok(), image_seq_len=3)
    assert p.expand_prompt(f"a {I}b") == f"a {F}{I * 3}{F} b"
    assert p.expand_prompt(f"{I}{I} x") == f"{F}{I * 3}{F}{I * 3}{F} x"
    ps = Idefics2Processor(Idefics2ImageProcessor(do_image_splitting=True), Tok(), image_seq_len=2)
    assert ps.expand_prompt(f"{I}") == F + (I * 2 + F) * 5
    with pytest.raises(ValueError):
        p(images=[np.zeros((60, 60, 3), np.uint8)], text=f"{I} and {I}")
    out = p(images=[np.zeros((60, 60, 3), np.uint8) + 9, np.zeros((80, 60, 3), np.uint8) + 9], text=[f"{I} one", f"two {I}"])
    assert out["pixel_values"].shape[:2] == (2, 1) and out["pixel_attention_mask"].shape[:2] == (2, 1)


def test_model_construction_weight_packing_and_image_bookkeeping_without_a_device():
    from tests.helpers import build_idefics2_model

    cfg = oi.tiny_cfg()
    W = oi.random_weights(cfg, seed=1, dtype=BF, **oi.TEST_WEIGHT_SCALES)
    model = build_idefics2_model(cfg, W, device="cpu", kv_pool_tokens=2048, max_seqs=4)
    lm, t, p = model.language_model, cfg.text, cfg.perceiver
    assert lm.head_dim == 128 and not lm.quantized
    assert tuple(lm._w["0.wqkv"].shape) == ((t.num_attention_heads + 2 * t.num_key_value_heads) * 128, t.hidden_size)
    assert torch.equal(lm._w["head"], W["language_model.lm_head.weight"])
    assert float(lm._w["0.bqkv"].abs().max()) == 0.0                                   # Mistral: no biases
    vt, cn = model.vision_model, model.connector
    assert vt.head_pad == 80 and vt._w["0.wqkv"].shape == (3 * 2 * 80, 144) and vt.side == 10
    assert cn._w["0.wq"].shape == (p.resampler_n_heads * 128, t.hidden_size)
    assert cn._w["0.wkv"].shape == (2 * p.num_key_value_heads * 128, t.hidden_size)
    gu = cn._w["mp_gu"]
    assert torch.equal(gu[0::2], W["connector.modality_projection.gate_proj.weight"]) and \
        torch.equal(gu[1::2], W["connector.modality_projection.up_proj.weight"])
    # padding images (all zero) are dropped, patch masks follow the pixel masks; position ids as the oracle's
    rng = np.random.default_rng(0)
    pv, pm = oi.preprocess([[rng.integers(1, 256, (90, 60, 3), dtype=np.uint8), rng.integers(1, 256, (56, 70, 3), dtype=np.uint8)],
                            [rng.integers(1, 256, (70, 70, 3), dtype=np.uint8)]], shortest_edge=56, longest_edge=140)
    assert pv.shape[:2] == (2, 2)
    real, pmask = model._real_images(pv, pm)
    oreal, opmask = oi.real_images_and_patch_mask(torch.from_numpy(pv), pm, cfg.vision.patch_size)
    assert real.shape[0] == 3 and np.array_equal(real, oreal.numpy()) and np.array_equal(pmask, opmask)
    from mlx_vlm_amd.models.idefics2.vision import bucket_position_ids
    assert np.array_equal(bucket_position_ids(pmask, vt.side), oi.position_ids(opmask, cfg.vision.num_patches_per_side))


def test_load_processor_and_prepare_inputs(tmp_path):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from mlx_vlm_amd import utils
    from mlx_vlm_amd.models.idefics2 import Idefics2Processor
    from tests.helpers import idefics2_config_from_oracle

    words = "what is in these pictures please".split()
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2, **{w: i + 3 for i, w in enumerate(words)}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>")
    fast.add_special_tokens({"additional_special_tokens": ["<fake_token_around_image>", "<image>", "<end_of_utterance>"]})
    fast.save_pretrained(str(tmp_path))
    (tmp_path / "preprocessor_config.json").write_text(json.dumps({"size": {"shortest_edge": 56, "longest_edge": 140},
                                                                   "do_image_splitting": False}))
    (tmp_path / "processor_config.json").write_text(json.dumps({"image_seq_len": 8}))
    cfg = idefics2_config_from_oracle(oi.tiny_cfg())
    proc = utils.load_processor(str(tmp_path), cfg)
    assert isinstance(proc, Idefics2Processor) and proc.image_seq_len == 8 and proc.image_processor.size["longest_edge"] == 140
    img_id = fast.convert_tokens_to_ids("<image>")
    rng = np.random.default_rng(0)
    ims = [rng.integers(0, 256, (90, 60, 3), dtype=np.uint8), rng.integers(0, 256, (56, 70, 3), dtype=np.uint8)]
    out = utils.prepare_inputs(proc, images=ims, prompts="what is <image> in these <image> pictures")
    assert int((out["input_ids"] == img_id).sum()) == 16
    assert out["pixel_values"].shape == (1, 2, 3, 90, 70) and out["pixel_attention_mask"].shape == (1, 2, 90, 70)
