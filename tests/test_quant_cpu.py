"""CPU tests of the MLX affine 4-bit path: the oracle's restatement (oracle/quant.py - "parity unpinned": no mlx in this
image, the vectors below are worked by hand from the published algorithm), the loader-side packing
(mlx-vlm_amd/models/quantized.py) and a 4-bit checkpoint through read_sanitized_weights -> Model.load_weights
without a device (reference utils.py:918-967)."""
import json

import numpy as np
import pytest
import torch

from oracle import qwen2_vl as oq
from oracle import quant as Q

BF = torch.bfloat16


def test_quantize_affine_hand_worked_group():
    """group = 0, 1, .., 15 repeated: w_min 0, w_max 15 -> |w_min| > |w_max| is false, so the scale is negative and the
    edge is w_max: scale = -(15 - 0) / 15 = -1, q0 = round(15 / -1) = -15, scale = 15 / -15 = -1, bias = 15,
    q = round((w - 15) / -1) = 15 - w.  Packing: element k of a row in word k // 8, nibble k % 8, little end first."""
    w = (torch.arange(64) % 16).to(BF)[None]
    wq, s, b = Q.quantize_affine(w)
    assert wq.shape == (1, 8) and s.shape == b.shape == (1, 1)
    assert float(s) == -1.0 and float(b) == 15.0
    q = 15 - (np.arange(64) % 16)
    words = [sum(int(q[8 * i + j]) << (4 * j) for j in range(8)) for i in range(8)]
    assert [int(x) & 0xFFFFFFFF for x in wq[0]] == words
    assert words[0] == 0x89ABCDEF and words[1] == 0x01234567
    assert torch.equal(Q.unpack(wq)[0], torch.from_numpy(q))
    assert torch.equal(Q.dequantize(wq, s, b), w)                      # integers: exact round trip
    # the mirrored group (0 .. -15): |w_min| > |w_max| -> positive scale, edge = w_min = -15, q = 15 + w ... = w / 1 + 15
    wq2, s2, b2 = Q.quantize_affine(-w)
    assert float(s2) == 1.0 and float(b2) == -15.0
    assert torch.equal(Q.dequantize(wq2, s2, b2), -w)
    # a constant-zero group: scale clamps at 1e-7 (sign flipped), q0 == 0 -> bias 0
    wq3, s3, b3 = Q.quantize_affine(torch.zeros(1, 64, dtype=BF))
    assert float(b3) == 0.0 and int(wq3.abs().sum()) == 0


def test_quantize_affine_error_bound_and_linear():
    """|w - dequant(quant(w))| <= one step |scale| (+ the bf16 rounding of scale / bias) on every element - half a step
    inside the grid, up to a step at the far end, which the edge-anchored scale may clip - and <= half a step for 90 % of
    them; quantized_linear == x . dequant^T in fp32."""
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(48, 256, generator=g) * 0.05).to(BF)
    wq, s, b = Q.quantize_affine(w)
    d32 = Q.dequantize(wq, s, b, dtype=torch.float32)
    err = (d32 - w.float()).abs().reshape(48, 4, 64)
    slack = 2 ** -8 * (15 * s.float().abs() + b.float().abs())[..., None]                       # bf16 scale x 15 steps, bias
    step = s.float().abs()[..., None]
    assert bool((err <= step + slack).all()), float((err - step - slack).max())
    assert float((err <= 0.5 * step + slack).float().mean()) > 0.9
    x = torch.randn(3, 256, generator=g).to(BF)
    y = Q.quantized_linear(x, wq, s, b)
    assert y.dtype == BF and torch.equal(y, (x.float() @ d32.T).to(BF))
    bias = torch.randn(48, generator=g).to(BF)
    assert torch.equal(Q.quantized_linear(x, wq, s, b, bias), (y.float() + bias.float()).to(BF))   # second typed op
    qw = Q.QW(wq, s, b)
    assert qw.shape == (48, 256) and torch.equal(qw.rows([3, 3, 0]), Q.dequantize(wq, s, b)[[3, 3, 0]])


def test_loader_packing_take_cat_interleave():
    from mlx_vlm_amd.models import quantized as Qz

    g = torch.Generator().manual_seed(1)
    w = (torch.randn(16, 128, generator=g) * 0.1).to(BF)
    wq, s, b = Q.quantize_affine(w)
    for words in (wq, wq.view(torch.uint32)):                    # what safetensors hands over for MLX's uint32
        q = Qz.take({"p.weight": words, "p.scales": s, "p.biases": b}, "p")
        assert q.wq.dtype == q.sb.dtype == torch.int32 and q.shape == (16, 128) and q.sb.shape == (16, 2)
        lo = (q.sb & 0xFFFF).to(torch.int16).view(BF)
        hi = ((q.sb >> 16) & 0xFFFF).to(torch.int16).view(BF)
        assert torch.equal(lo, s) and torch.equal(hi, b)
        assert torch.equal(q.wq, wq)
    a, c = q.rows(slice(0, 8)), q.rows(slice(8, 16))
    assert torch.equal(Qz.cat_rows([a, c]).wq, q.wq) and torch.equal(Qz.cat_rows([a, c]).sb, q.sb)
    il = Qz.interleave_rows(a, c)
    assert torch.equal(il.wq[0::2], a.wq) and torch.equal(il.sb[1::2], c.sb)
    with pytest.raises(ValueError):
        Qz.take({"p.weight": wq, "p.scales": s[:, :1], "p.biases": b[:, :1]}, "p")
    with pytest.raises(ValueError):
        Qz.take({"p.weight": w, "p.scales": s, "p.biases": b}, "p")
    Qz.check_quantization(None)
    Qz.check_quantization({"group_size": 64, "bits": 4})
    for bad in ({"group_size": 32, "bits": 4}, {"group_size": 64, "bits": 8}, {"group_size": 64, "bits": 4, "mode": "mxfp4"}):
        with pytest.raises(NotImplementedError):
            Qz.check_quantization(bad)


def test_4bit_checkpoint_reads_and_packs_without_a_device(tmp_path):
    """MLX-layout 4-bit checkpoint (language model quantized, tower bf16) -> read_sanitized_weights -> Model.load_weights
    on the host: the engine tables hold the packed words (q/k/v rows concatenated, gate/up interleaved, embedding rows)."""
    from safetensors.torch import save_file

    from mlx_vlm_amd import utils
    from mlx_vlm_amd.models import quantized as Qz
    from mlx_vlm_amd.models.qwen2_vl import Model
    from tests.helpers import model_config_from_oracle

    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=7, dtype=BF, std=0.05, embed_std=0.2)
    ck, ow = Q.quantize_checkpoint(W, predicate=lambda p, v: p.startswith("language_model."))
    assert "language_model.model.embed_tokens.scales" in ck and "vision_tower.blocks.0.attn.qkv.scales" not in ck
    save_file({k: (v.view(torch.uint32) if v.dtype == torch.int32 else v).contiguous() for k, v in ck.items()},
              str(tmp_path / "model.safetensors"))
    model = Model(model_config_from_oracle(cfg), device="cpu", kv_pool_tokens=1024, max_seqs=2)
    got = utils.read_sanitized_weights(str(tmp_path), model, {"quantization": {"group_size": 64, "bits": 4}})
    assert got["language_model.model.layers.0.mlp.down_proj.weight"].dtype in (torch.uint32, torch.int32)
    with pytest.raises(NotImplementedError):
        utils.read_sanitized_weights(str(tmp_path), model, {"quantization": {"group_size": 128, "bits": 4}})
    model.load_weights(got)
    lm = model.language_model
    assert lm.quantized
    t = cfg.text
    hd = t.hidden_size // t.num_attention_heads
    wqkv = lm._w["0.wqkv"]
    assert isinstance(wqkv, Qz.QuantW) and wqkv.shape == ((t.num_attention_heads + 2 * t.num_key_value_heads) * hd, t.hidden_size)
    qk = ow["language_model.model.layers.0.self_attn.k_proj.weight"]
    nq = t.num_attention_heads * hd
    assert torch.equal(wqkv.wq[nq:nq + qk.wq.shape[0]], qk.wq)
    wgu = lm._w["0.wgu"]
    assert torch.equal(wgu.wq[1::2], ow["language_model.model.layers.0.mlp.up_proj.weight"].wq)
    emb = lm._w["embed"]
    assert isinstance(emb, Qz.QuantW) and emb.shape == (t.vocab_size, t.hidden_size)
    # a half-quantized projection group is refused
    bad = dict(got)
    k = "language_model.model.layers.1.self_attn.v_proj"
    bad[k + ".weight"] = W[k + ".weight"]
    del bad[k + ".scales"], bad[k + ".biases"]
    with pytest.raises((NotImplementedError, ValueError)):
        Model(model_config_from_oracle(cfg), device="cpu", kv_pool_tokens=1024, max_seqs=2).load_weights(bad)
