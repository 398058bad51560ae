"""Phi-3.5-vision host logic on the CPU (no kernel is launched): the 96 -> 128 head layouts of models/phi3_v/language.py
reproduce the reference attention exactly (bf16 and MLX 4-bit), Su-scaled RoPE as frequency table + attention scale,
model construction and weight packing without a device, processor / prepare_inputs / load plumbing."""
import json

import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import phi3_v as op
from oracle import quant as Q

BF = torch.bfloat16


def _lm_stub(hd=96):
    from mlx_vlm_amd.models.phi3_v.language import LanguageModel

    lm = object.__new__(LanguageModel)
    lm.real_head_dim = hd
    return lm


def _attend(x, wq, wk, wv, wo, H, width, inv, scale, pos, mscale=None):
    B, L, _ = x.shape
    q, k, v = (O.linear(x, w).reshape(B, L, H, width).permute(0, 2, 1, 3) for w in (wq, wk, wv))
    if mscale is not None:      # SuScaledRoPE: x * T(scale), a typed multiply, before the rotation
        q, k = (q.float() * mscale).to(q.dtype), (k.float() * mscale).to(k.dtype)
    q, k = O.mrope_apply(q, pos, inv, None, "fused"), O.mrope_apply(k, pos, inv, None, "fused")
    o = O.sdpa(q, k, v, scale=scale, causal=True)
    return O.linear(o.permute(0, 2, 1, 3).reshape(B, L, -1), wo)


def test_heads_96_in_128_columns_reproduce_rope_attention_exactly():
    """q / k rows with the rotary halves at columns 0 and 64, v rows / o_proj columns contiguous at 128 h + (96 h mod 64):
    rotate-half RoPE over 128 columns (48 real frequencies + zeros) and attention on that layout are bit-identical to the
    96-wide computation - the zero coordinates stay zero under rotation and add exact zeros to q.k and P.V."""
    from mlx_vlm_amd.models.phi3_v.language import ENGINE_HEAD_DIM as E

    torch.manual_seed(5)
    B, L, D, H, hd = 1, 11, 192, 3, 96
    x = torch.randn(B, L, D).to(BF)
    wq, wk, wv = (torch.randn(H * hd, D).mul(0.1).to(BF) for _ in range(3))
    wo = torch.randn(D, H * hd).mul(0.1).to(BF)
    pos = torch.arange(7, 7 + L)[None]
    short, _ = op.su_factors(hd)
    inv96 = 1.0 / (torch.tensor(short) * 10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ref = _attend(x, wq, wk, wv, wo, H, hd, inv96, hd ** -0.5, pos)
    lm = _lm_stub(hd)
    qi, vi = lm._qk_index(H), lm._v_index(H)
    assert [int((vi.reshape(H, E)[h] >= 0).argmax()) for h in range(H)] == [0, 32, 0]
    inv = torch.zeros(E // 2)
    inv[: hd // 2] = inv96
    got = _attend(x, lm._rows(wq, qi), lm._rows(wk, qi), lm._rows(wv, vi), lm._o_cols(wo, H), H, E, inv, hd ** -0.5, pos)
    assert torch.equal(got, ref)


def test_su_scale_rounding_matters_and_is_kept():
    """The reference multiplies q and k by T(scale) - rounded to bf16 - before rotating.  Folding scale ** 2 into the
    softmax scale instead (no rounding) is the same function in exact arithmetic but moves outputs by up to ~1 % of their
    scale here (attention amplifies the 2 ** -9 perturbation of q and k): the engine therefore applies the typed multiply
    itself (`rope_qk_scale`), and this test pins the reason."""
    from mlx_vlm_amd.models.phi3_v.language import su_scale

    torch.manual_seed(6)
    B, L, D, H, hd = 1, 13, 192, 2, 96
    x = torch.randn(B, L, D).to(BF)
    wq, wk, wv = (torch.randn(H * hd, D).mul(0.1).to(BF) for _ in range(3))
    wo = torch.randn(D, H * hd).mul(0.1).to(BF)
    pos = torch.arange(0, L)[None]
    t = op.tiny_cfg().text
    inv_s, _, s = op.su_rope_tables(t)
    assert s == su_scale(t.max_position_embeddings, t.original_max_position_embeddings) == 1.1875
    ref = _attend(x, wq, wk, wv, wo, H, hd, inv_s, hd ** -0.5, pos, mscale=s)
    folded = _attend(x, wq, wk, wv, wo, H, hd, inv_s, s * s * hd ** -0.5, pos)
    err = (folded.float() - ref.float()).abs()
    assert 2.0 ** -9 * float(ref.float().abs().max()) < float(err.max()) < 0.03 * float(ref.float().abs().max())


@pytest.mark.parametrize("H", [2, 4, 32])
def test_4bit_o_proj_columns_and_qkv_rows_move_without_requantization(H):
    """MLX 4-bit weights in the engine layout: every packed word and every (scale, bias) pair is MOVED - the
    dequantized engine matrix equals the dequantized checkpoint matrix at the mapped coordinates and is exactly zero in
    the zero rows; o_proj's padding columns may hold anything (they multiply exact zeros) but every real column is
    reproduced bit for bit."""
    from mlx_vlm_amd.models import quantized as Qz
    from mlx_vlm_amd.models.phi3_v.language import ENGINE_HEAD_DIM as E

    hd, D = 96, 128
    g = torch.Generator().manual_seed(H)
    wo = (torch.randn(D, H * hd, generator=g) * 0.1).to(BF)
    wqkv = (torch.randn(H * hd, D, generator=g) * 0.1).to(BF)
    ck, _ = Q.quantize_checkpoint({"o.weight": wo, "q.weight": wqkv})
    lm = _lm_stub(hd)
    qo, qq = Qz.take(ck, "o"), Qz.take(ck, "q")

    def deq(qw):
        sc = (qw.sb & 0xFFFF).to(torch.int16).view(BF)
        bi = ((qw.sb >> 16) & 0xFFFF).to(torch.int16).view(BF)
        return Q.dequantize(qw.wq, sc, bi, dtype=torch.float32)

    full_o, full_q = deq(qo), deq(qq)
    vi, qi = lm._v_index(H), lm._qk_index(H)
    eng_o = deq(lm._o_cols(qo, H))
    assert eng_o.shape == (D, H * E)
    assert torch.equal(eng_o[:, vi >= 0], full_o[:, vi[vi >= 0]])
    eng_q = deq(lm._rows(qq, qi))
    assert torch.equal(eng_q[qi >= 0], full_q[qi[qi >= 0]]) and float(eng_q[qi < 0].abs().max()) == 0.0
    eng_v = deq(lm._rows(qq, vi))
    assert torch.equal(eng_v[vi >= 0], full_q[vi[vi >= 0]]) and float(eng_v[vi < 0].abs().max()) == 0.0


@pytest.mark.parametrize("w4", [False, True])
def test_model_construction_and_weight_packing_run_without_a_device(w4):
    from mlx_vlm_amd.models import quantized as Qz
    from tests.helpers import build_phi3v_model

    cfg = op.tiny_cfg()
    W = op.random_weights(cfg, seed=1, dtype=BF, **op.TEST_WEIGHT_SCALES)
    ck = W
    if w4:
        ck, _ = Q.quantize_checkpoint(W, predicate=lambda p, v: not p.startswith("model.vision_embed_tokens."))
    model = build_phi3v_model(cfg, ck, device="cpu", kv_pool_tokens=2048, max_seqs=4)
    lm, t = model.language_model, cfg.text
    assert lm.head_dim == 128 and lm.pool.head_dim == 128 and lm.quantized == w4
    H = t.num_attention_heads
    shape = lambda w: tuple(w.shape)   # noqa: E731
    assert shape(lm._w["0.wqkv"]) == (3 * H * 128, t.hidden_size) and shape(lm._w["0.wo"]) == (t.hidden_size, H * 128)
    assert shape(lm._w["0.wgu"]) == (2 * t.intermediate_size, t.hidden_size)
    if not w4:
        gu = W["model.layers.0.mlp.gate_up_proj.weight"]
        assert torch.equal(lm._w["0.wgu"][0::2], gu[: t.intermediate_size]) and torch.equal(lm._w["0.wgu"][1::2], gu[t.intermediate_size:])
        assert torch.equal(lm._w["head"], W["lm_head.weight"])
    else:
        assert isinstance(lm._w["0.wo"], Qz.QuantW) and isinstance(lm._w["embed"], Qz.QuantW)
    inv = lm._w["inv_freq"]
    short = torch.tensor(t.short_factor)
    want = 1.0 / (short * t.rope_theta ** (torch.arange(0, 96, 2).float() / 96))
    assert torch.allclose(inv[:48], want, rtol=1e-6) and float(inv[48:64].abs().max()) == 0.0
    # both regimes are resident: the long-factor table sits behind the short one (vlm_llm_config.rope_long_from)
    want_long = 1.0 / (torch.tensor(t.long_factor) * t.rope_theta ** (torch.arange(0, 96, 2).float() / 96))
    assert inv.numel() == 128 and torch.allclose(inv[64:112], want_long, rtol=1e-6) and float(inv[112:].abs().max()) == 0.0
    assert lm.rope_long_from == t.original_max_position_embeddings
    assert abs(lm.args.attn_scale - 96 ** -0.5) < 1e-7 and lm.args.rope_qk_scale == 1.1875
    vt = model.vision_model
    assert vt.n_run_layers == cfg.vision.num_hidden_layers - 1 and f"{vt.n_run_layers}.wqkv" not in vt._w
    assert vt._w["0.wqkv"].shape == (3 * 1024, 1024) and vt._w["wpatch"].shape == (1024, 640)
    # SuScaledRoPE's per-call rule (rope_utils.py:168-172): max cache offset + tokens of the call > original_max
    from types import SimpleNamespace as NS
    mk = lambda off: [NS(_seq=NS(offset=off))]   # noqa: E731
    lim = t.original_max_position_embeddings
    assert lm._call_regime([mk(0)], [lim]) is False and lm._call_regime([mk(0)], [lim + 1]) is True
    assert lm._call_regime([mk(0), mk(lim - 3)], [2, 4]) is True and lm._call_regime([mk(lim - 5), mk(7)], [2, 5]) is False


def test_processor_bit_exact_and_token_rule():
    """Phi3VImageProcessor of the product (table-driven) vs the oracle's restatement of the reference's arithmetic."""
    from mlx_vlm_amd.models.phi3_v import Phi3VImageProcessor

    rng = np.random.default_rng(3)
    ip = Phi3VImageProcessor()
    for hw in ((336, 336), (100, 333), (500, 120), (37, 41)):
        im = rng.integers(0, 256, (*hw, 3), dtype=np.uint8)
        out = ip([im])
        pv, sz = op.preprocess([im])
        assert out["pixel_values"].dtype == np.float32 and np.array_equal(out["pixel_values"], pv)
        assert np.array_equal(out["image_sizes"], sz)
        assert ip.calc_num_image_tokens(im) == op.num_image_tokens(hw[1], hw[0])
    a, b = rng.integers(0, 256, (336, 336, 3), dtype=np.uint8), rng.integers(0, 256, (300, 90, 3), dtype=np.uint8)
    both = ip([a, b])
    pv, sz = op.preprocess([a, b])
    assert np.array_equal(both["pixel_values"], pv) and np.array_equal(both["image_sizes"], sz)
    assert Phi3VImageProcessor(num_crops=16).calc_num_image_tokens(a) == op.num_image_tokens(336, 336, num_crops=16)


def test_load_processor_and_prepare_inputs(tmp_path):
    """load_processor picks Phi3VProcessor for model_type phi3_v (preprocessor_config num_crops honoured); prepare_inputs
    returns input_ids with negative image runs, pixel_values [1, T, 3, 336, 336] and image_sizes."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from mlx_vlm_amd import utils
    from mlx_vlm_amd.models.phi3_v import Phi3VProcessor
    from tests.helpers import phi3v_config_from_oracle

    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2, **{w: i + 3 for i, w in enumerate("what is in this picture please tell me".split())}}
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    PreTrainedTokenizerFast(tokenizer_object=tk, unk_token="<unk>", bos_token="<s>", eos_token="</s>").save_pretrained(str(tmp_path))
    (tmp_path / "preprocessor_config.json").write_text(json.dumps({"num_crops": 16, "num_img_tokens": 144}))
    cfg = phi3v_config_from_oracle(op.tiny_cfg())
    proc = utils.load_processor(str(tmp_path), cfg)
    assert isinstance(proc, Phi3VProcessor) and proc.image_processor.num_crops == 16
    assert proc.tokenizer.stopping_criteria is not None and proc.detokenizer is not None
    im = np.random.default_rng(0).integers(0, 256, (336, 336, 3), dtype=np.uint8)
    out = utils.prepare_inputs(proc, images=[im], prompts="what is <|image_1|> in this picture")
    n = proc.image_processor.calc_num_image_tokens(im)
    ids = out["input_ids"]
    assert ids.shape[0] == 1 and int((ids == -1).sum()) == n == op.num_image_tokens(336, 336, num_crops=16)
    assert out["pixel_values"].shape == (1, 17, 3, 336, 336) and out["image_sizes"].tolist() == [[1344, 1344]]
    text_only = utils.prepare_inputs(proc, images=None, prompts="what is this")
    assert "pixel_values" not in text_only and text_only["input_ids"].shape == (1, 3)
