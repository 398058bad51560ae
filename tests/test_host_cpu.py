"""CPU tests of the host side of the product package: C-ABI library loads and exports every symbol of
include/vlm_hip.h (no compute calls), host logic (rope index, merge rows, processor, sanitize, page
allocator, request sharding) against the oracle / goldens, and the N > 1 path on gloo with world_size 2."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import image_processor as oip
from oracle import qwen2_vl as oq
from tests.helpers import model_config_from_oracle, synth_request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "qwen2_vl_tiny_hf.npz"))


def test_library_exports_every_declared_symbol():
    from mlx_vlm_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge

        ge.build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "vlm_hip.h")).read()
    declared = set(re.findall(r"\b(vlm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES)
    for s in declared:
        assert hasattr(L, s), s
    assert L.vlm_abi_version() == 8
    # ... and NOTHING else: the dynamic symbol table of the .so is exactly the header (debug hooks and library-internal
    # entry points have hidden visibility)
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_no_cpu_fallback_ops_raise_on_cpu_tensors():
    from mlx_vlm_amd import _lib, ops

    with pytest.raises(_lib.VlmHipError):
        ops.gemm(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mlx-vlm_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dp, f)


@pytest.fixture(scope="module")
def lm():
    from mlx_vlm_amd.models.qwen2_vl import LanguageModel

    cfg = oq.tiny_cfg()
    mc = model_config_from_oracle(cfg)
    return cfg, LanguageModel(mc.text_config, mc, device="cpu")


@pytest.mark.parametrize("case", ["one_image", "two_images"])
def test_get_rope_index_vs_hf_golden(lm, case):
    _, m = lm
    pos, deltas = m.get_rope_index(G[case + ".input_ids"], G[case + ".grid_thw"])
    np.testing.assert_array_equal(pos, G[case + ".hf_position_ids"])
    np.testing.assert_array_equal(deltas, G[case + ".hf_rope_deltas"])


def test_get_rope_index_vs_oracle_random_cases(lm):
    cfg, m = lm
    rng = np.random.default_rng(0)
    for trial in range(20):
        n_img = int(rng.integers(0, 4))
        sizes = [(28 * int(rng.integers(2, 6)), 28 * int(rng.integers(2, 6))) for _ in range(n_img)]
        if n_img:
            ids, _, thw = synth_request(cfg, sizes, n_text=int(rng.integers(1, 9)), seed=trial)
            # interleave some text between images
            a, b = m.get_rope_index(ids, thw), oq.get_rope_index(cfg, ids, thw)
        else:
            ids = rng.integers(3, 900, (2, 9))
            mask = np.ones_like(ids)
            mask[1, :3] = 0
            a, b = m.get_rope_index(ids, attention_mask=mask), oq.get_rope_index(cfg, ids, attention_mask=mask)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])


def test_get_rope_index_left_padded_batch_with_images(lm):
    cfg, m = lm
    ids1, _, thw1 = synth_request(cfg, [(56, 56)], n_text=5, seed=1)
    ids2, _, thw2 = synth_request(cfg, [(56, 84)], n_text=9, seed=2)
    L = max(ids1.shape[1], ids2.shape[1])
    ids = np.full((2, L), 2, dtype=np.int64)
    mask = np.zeros((2, L), dtype=np.int64)
    for r, x in enumerate((ids1, ids2)):
        ids[r, L - x.shape[1]:] = x[0]
        mask[r, L - x.shape[1]:] = 1
    thw = np.concatenate([thw1, thw2])
    a, b = m.get_rope_index(ids, thw, None, mask), oq.get_rope_index(cfg, ids, thw, None, mask)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_processor_matches_oracle_and_placeholder_expansion():
    from mlx_vlm_amd.models.qwen2_vl.processing_qwen2_vl import Qwen2VLImageProcessor, Qwen2VLProcessor, smart_resize

    for h, w, rh, rw in G["smart_resize.table"].tolist():
        assert smart_resize(h, w) == (rh, rw)
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (3, 100, 150), dtype=np.uint8), rng.integers(0, 256, (3, 336, 336), dtype=np.uint8)]
    out = Qwen2VLImageProcessor()(imgs)
    pix, thw = oip.process(imgs)
    np.testing.assert_array_equal(out["pixel_values"], pix)
    np.testing.assert_array_equal(out["image_grid_thw"], thw)

    class Tok:  # reference tests use a mock tokenizer too (tests/test_processors.py:1310-1328)
        def __call__(self, text, **kw):
            return {"input_ids": [[len(t.split("<|image_pad|>")) - 1 for t in text]]}

    p = Qwen2VLProcessor(Qwen2VLImageProcessor(), Tok())
    o = p(images=imgs[:1], text="a<|image_pad|>b")
    assert o["input_ids"][0][0] == int(thw[0].prod()) // 4       # placeholder expanded to grid.prod() // 4 copies


def test_sanitize_hf_layouts():
    from mlx_vlm_amd.models.qwen2_vl import Model
    from mlx_vlm_amd.models.qwen2_vl.vision import VisionModel

    keys = ["visual.blocks.0.attn.qkv.weight", "model.layers.0.mlp.up_proj.weight", "lm_head.weight",
            "model.visual.merger.mlp.0.bias", "model.language_model.norm.weight"]
    out = Model.sanitize(None, {k: 0 for k in keys})
    assert set(out) == {"vision_tower.blocks.0.attn.qkv.weight", "language_model.model.layers.0.mlp.up_proj.weight",
                        "language_model.lm_head.weight", "vision_tower.merger.mlp.0.bias", "language_model.model.norm.weight"}
    w = torch.zeros(32, 3, 2, 14, 14)
    v = VisionModel.sanitize(None, {"patch_embed.proj.weight": w, "x.position_ids": 1})
    assert v["patch_embed.proj.weight"].shape == (32, 2, 14, 14, 3) and "x.position_ids" not in v


def test_model_config_from_dict_root_level_text_config():
    from mlx_vlm_amd import synthetic
    from mlx_vlm_amd.models.qwen2_vl import ModelConfig

    c = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))
    assert c.text_config.hidden_size == 1536 and c.text_config.num_key_value_heads == 2
    assert c.vision_config.embed_dim == 1280 and c.image_token_id == 151655
    with pytest.raises(ValueError):
        ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B, rope_scaling={"type": "linear", "mrope_section": [1, 1, 1]}))


def test_kv_pool_allocator_cpu():
    from mlx_vlm_amd.models.cache import KVCache, KVPool, PagedSequence

    pool = KVPool(2, 1, 128, max_tokens=64 * 6, max_seqs=4, device="cpu", layout="paged")
    a, b = PagedSequence(pool), PagedSequence(pool)
    assert (a.seq, b.seq) == (0, 1)
    a.reserve(65)
    b.reserve(1)
    assert len(a.pages) == 2 and len(b.pages) == 1 and len(set(a.pages + b.pages)) == 3
    assert pool.block_table[0, :2].tolist() == a.pages
    with pytest.raises(RuntimeError):
        b.reserve(64 * 5)                      # out of pages
    a.release()
    assert pool.new_seqs(2) == [2, 3]           # row 0 freed, row 1 in use -> the first run of two consecutive rows
    c = [KVCache(b, i) for i in range(2)]
    b.offset = 10
    assert [x.trim(3) for x in c] == [3, 3] and b.offset == 7     # trims once (views share the sequence)
    assert c[0].state[0].shape == (1, 1, 7, 128)


def test_kv_pool_identity_layout_cpu():
    """identity layout: slot s owns pages [s*max_pages, (s+1)*max_pages); the block table is still filled."""
    from mlx_vlm_amd.models.cache import KVPool, PagedSequence

    pool = KVPool(2, 1, 128, max_tokens=64 * 6, max_seqs=4, device="cpu")          # "auto" -> identity (tiny)
    assert pool.identity and pool.n_pages == 4 * pool.max_pages
    a, b = PagedSequence(pool), PagedSequence(pool)
    a.reserve(130)
    b.reserve(64 * 6)
    assert a.pages == [0, 1, 2] and b.pages == [pool.max_pages + i for i in range(6)]
    assert pool.block_table[1, :6].tolist() == b.pages
    with pytest.raises(RuntimeError):
        b.reserve(64 * 6 + 1)                   # beyond max_pages_per_seq
    a.release()
    c = PagedSequence(pool)
    c.reserve(1)
    assert c.seq == 0 and c.pages == [0]        # the slot's region is reused, never another slot's


def test_shard_requests_partition():
    from mlx_vlm_amd.parallel import shard_requests

    lens = [5, 100, 7, 50, 60, 9, 80, 3, 41]
    parts = [shard_requests(len(lens), r, 4, lens) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(len(lens)))
    sums = [sum(lens[i] for i in p) for p in parts]
    assert max(sums) - min(sums) <= max(lens)


GLOO_WORKER = r"""
import os, sys, torch
sys.path.insert(0, %r)
from mlx_vlm_amd import parallel
rank, ws, local = parallel.init(backend="gloo")
assert ws == 2
g = torch.Generator().manual_seed(0)
W = {f"w{i}": (torch.randn(33 + i, 17, generator=g) if rank == 0 else torch.zeros(33 + i, 17)).to(torch.bfloat16) for i in range(5)}
W["ids"] = torch.arange(10) if rank == 0 else torch.zeros(10, dtype=torch.long)
parallel.broadcast_weights(W, src=0, bucket_bytes=2048)
g = torch.Generator().manual_seed(0)
for i in range(5):
    assert torch.equal(W[f"w{i}"], torch.randn(33 + i, 17, generator=g).to(torch.bfloat16)), i
assert W["ids"].tolist() == list(range(10))
mine = parallel.shard_requests(7, rank, ws)
res = parallel.gather_results([(i, i * i) for i in mine])
assert abs(parallel.max_over_ranks(float(rank)) - 1.0) < 1e-9 and abs(parallel.sum_over_ranks(1.0) - 2.0) < 1e-9
assert parallel.per_rank(10.0 + rank) == [10.0, 11.0]
parallel.barrier()
if rank == 0:
    flat = sorted(sum(res, []))
    assert flat == [(i, i * i) for i in range(7)], flat
    print("GLOO_OK")
"""


def test_dp_path_gloo_world_size_2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(GLOO_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "GLOO_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


DP_WORKER = r"""
import os, sys, json, zlib
import numpy as np, torch
sys.path.insert(0, %r)
from mlx_vlm_amd import parallel
from oracle import qwen2_vl as oq
from tests.test_engine_gpu import _write_tiny_checkpoint
import pathlib
rank, ws, local = parallel.init(backend="gloo")
ckpt = pathlib.Path(os.environ["DP_CKPT"])
cfg = oq.tiny_cfg()
if rank == 0:
    W = oq.random_weights(cfg, seed=1234, dtype=torch.bfloat16, std=0.05, embed_std=0.2)
    _write_tiny_checkpoint(ckpt, cfg, W)
parallel.barrier()
if rank == 1:                       # rank 1 must not need the weight file: its directory holds config + tokenizer and a
    import shutil                   # safetensors file the loader would choke on
    mine = ckpt.parent / "ckpt_rank1"
    shutil.copytree(ckpt, mine)
    (mine / "model.safetensors").write_bytes(b"")
    ckpt = mine
parallel.barrier()
model, processor, stats = parallel.dp_load(str(ckpt), device="cpu", bucket_bytes=1 << 16, kv_pool_tokens=1024, max_seqs=4)
assert stats["ranks"] == 2 and stats["weight_bytes"] > 0 and stats["broadcast_s"] > 0
# every rank holds the same packed replica
lm = model.language_model
crc = zlib.crc32(lm._w["0.wqkv"].view(torch.int16).numpy().tobytes()) ^ zlib.crc32(lm._w["embed"].view(torch.int16).numpy().tobytes())
allc = parallel.gather_results([crc])
# requests of different lengths; the mock engine "generates" a function of the prompt, so the gathered result can be
# compared with the single-process answer
rng = np.random.default_rng(0)
reqs = [{"input_ids": rng.integers(3, 1000, int(n)), "max_tokens": 3 + i %% 4} for i, n in enumerate(rng.integers(4, 40, 11))]
def mock(indices, rq, mt):
    return [[int(rq[i]["input_ids"].sum() %% 997) + k for k in range(mt[i])] for i in indices]
served = []
def mock_logged(indices, rq, mt):
    served.extend(indices)
    return mock(indices, rq, mt)
out = parallel.dp_batch_generate(model, None, requests=reqs, serve=mock_logged)
lens = [len(r["input_ids"]) for r in reqs]
assert served == parallel.shard_requests(len(reqs), rank, ws, lens)
# prompts= form: a rank tokenises ONLY the requests it is dealt (the deal sorts by a tokeniser-free length proxy)
from mlx_vlm_amd import utils as U
prompts = ["w%%d " %% i * (3 + (7 * i) %% 11) for i in range(9)]
prepared = []
def fake_prepare(processor, images=None, prompts=None, **kw):
    prepared.append(prompts)
    return {"input_ids": np.arange(3, 3 + len(prompts.split()))[None]}
U.prepare_inputs = fake_prepare
served2 = []
def mock2(indices, rq, mt):
    served2.extend(indices)
    assert all(rq[i] is not None for i in indices)
    return [[int(rq[i]["input_ids"].sum() %% 997)] * mt[i] for i in indices]
out2 = parallel.dp_batch_generate(model, None, prompts=prompts, serve=mock2, max_tokens=2)
mine2 = parallel.shard_requests(len(prompts), rank, ws, [len(p) for p in prompts])
assert served2 == mine2 and prepared == [prompts[i] for i in mine2], (served2, mine2, prepared)
if rank == 0:
    assert len(set(allc[0] + allc[1])) == 1, allc
    assert out["tokens"] == mock(list(range(len(reqs))), reqs, [r["max_tokens"] for r in reqs])
    assert out["ranks"] == 2 and sorted(out["per_rank_requests"]) == [5, 6]
    assert [len(t) for t in out2["tokens"]] == [2] * 9 and sorted(out2["per_rank_requests"]) == [4, 5]
    assert out2["tokens"] == [[int(np.arange(3, 3 + len(p.split())).sum() %% 997)] * 2 for p in prompts]
    print("DP_OK")
else:
    assert out is None and out2 is None
parallel.shutdown()
"""


def test_dp_load_and_dp_batch_generate_gloo_world_size_2(tmp_path):
    """The product entry points of the data-parallel path end to end on 2 gloo ranks: `dp_load` (rank 0 reads the
    checkpoint, the replica is broadcast in buckets, rank 1 never opens the file) and `dp_batch_generate` (length-sorted
    deal, per-rank serving, gather in the original order) against the single-process answer of a mock engine."""
    script = tmp_path / "w.py"
    script.write_text(DP_WORKER % ROOT)
    (tmp_path / "ckpt").mkdir()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29614", DP_CKPT=str(tmp_path / "ckpt"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29614", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "DP_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_load_errors_match_reference_contract(tmp_path):
    """FileNotFoundError without a local directory / safetensors (reference utils.py:781-801,1209-1210), ValueError for an
    unknown model_type (utils.py:633-635)."""
    import json

    from mlx_vlm_amd import load

    with pytest.raises(FileNotFoundError):
        load(str(tmp_path / "missing"))
    (tmp_path / "config.json").write_text(json.dumps(dict(model_type="qwen2_vl")))
    with pytest.raises(FileNotFoundError):
        load(str(tmp_path))
    from safetensors.torch import save_file

    save_file({"x": torch.zeros(1)}, str(tmp_path / "model.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps(dict(model_type="not_a_model")))
    with pytest.raises(ValueError):
        load(str(tmp_path))


def test_package_level_names_are_callable_in_any_import_order():
    """`from mlx_vlm_amd import generate` after the submodule was imported (load() does that) must still be callable."""
    import importlib

    import mlx_vlm_amd

    importlib.import_module("mlx_vlm_amd.generate")
    importlib.import_module("mlx_vlm_amd.utils")
    for name in ("load", "generate", "stream_generate", "batch_generate", "generate_step", "prepare_inputs"):
        assert callable(getattr(mlx_vlm_amd, name)), name
    from mlx_vlm_amd.generate import generate as fn

    assert fn.__name__ == "generate"


# ---- nanoLLaVA (llava_bunny): the head-width mappings onto the engine's kernels are exact (no GPU needed)
def test_bunny_llm_heads_spread_64_to_128_reproduce_rope_attention_exactly():
    """language.py of models/llava_bunny lays 64-wide heads out in 128 columns (real dims [0,32) -> [0,32), [32,64) ->
    [64,96)), gives the engine a frequency table with 32 real entries + zeros and attn_scale = 64 ** -0.5.  Emulated with
    the oracle's primitives: q/k/v projections, rotate-half RoPE over 128 columns and attention on the spread layout
    give bit-identical outputs (after o_proj) to the 64-wide computation - zeros stay zeros under rotation and add
    exact zeros to every dot product."""
    from mlx_vlm_amd.models.llava_bunny.language import ENGINE_HEAD_DIM, LanguageModel
    from oracle import ops as O

    torch.manual_seed(3)
    B, L, D, H, hd = 1, 9, 128, 2, 64
    bf = torch.bfloat16
    x = torch.randn(B, L, D).to(bf)
    wq, wk, wv = (torch.randn(H * hd, D).mul(0.1).to(bf) for _ in range(3))
    bq, bk, bv = (torch.randn(H * hd).mul(0.1).to(bf) for _ in range(3))
    wo = torch.randn(D, H * hd).mul(0.1).to(bf)
    pos = torch.arange(5, 5 + L)[None]

    def attend(wq, bq, wk, bk, wv, bv, wo, width, inv, scale):
        q, k, v = (O.linear(x, w, b).reshape(B, L, H, width).permute(0, 2, 1, 3) for w, b in ((wq, bq), (wk, bk), (wv, bv)))
        q, k = O.mrope_apply(q, pos, inv, None, "fused"), O.mrope_apply(k, pos, inv, None, "fused")
        o = O.sdpa(q, k, v, scale=scale, causal=True)
        return O.linear(o.permute(0, 2, 1, 3).reshape(B, L, -1), wo)

    ref = attend(wq, bq, wk, bk, wv, bv, wo, hd, O.mrope_inv_freq(hd, 1e6), hd ** -0.5)
    lm = object.__new__(LanguageModel)
    lm.real_head_dim = hd
    inv = torch.zeros(ENGINE_HEAD_DIM // 2)
    inv[: hd // 2] = O.mrope_inv_freq(hd, 1e6)
    got = attend(lm._spread(wq, H), lm._spread(bq, H), lm._spread(wk, H), lm._spread(bk, H), lm._spread(wv, H),
                 lm._spread(bv, H), lm._spread(wo.t(), H).t().contiguous(), ENGINE_HEAD_DIM, inv, hd ** -0.5)
    assert torch.equal(got, ref)


def test_bunny_siglip_heads_padded_72_to_80_reproduce_attention_exactly():
    """vision.py of models/llava_bunny zero-pads every 72-wide SigLIP head to the 80 the attention kernel has."""
    from oracle import ops as O

    torch.manual_seed(4)
    N, E, H, hd, hp = 37, 144, 2, 72, 80
    bf = torch.bfloat16
    x = torch.randn(1, N, E).to(bf)
    w = {n: torch.randn(E, E).mul(0.1).to(bf) for n in "qkvo"}
    b = {n: torch.randn(E).mul(0.1).to(bf) for n in "qkvo"}

    def heads(t, width):
        return t.reshape(1, N, H, width).permute(0, 2, 1, 3)

    q, k, v = (heads(O.linear(x, w[n], b[n]), hd) for n in "qkv")
    ref = O.linear(O.sdpa(q, k, v, scale=hd ** -0.5).permute(0, 2, 1, 3).reshape(1, N, E), w["o"], b["o"])

    def pad_rows(m):
        out = torch.zeros(H, hp, m.shape[1], dtype=bf)
        out[:, :hd] = m.reshape(H, hd, -1)
        return out.reshape(H * hp, -1)

    def pad_vec(vv):
        out = torch.zeros(H, hp, dtype=bf)
        out[:, :hd] = vv.reshape(H, hd)
        return out.reshape(-1)

    qp, kp, vp = (heads(O.linear(x, pad_rows(w[n]), pad_vec(b[n])), hp) for n in "qkv")
    wo = torch.zeros(E, H, hp, dtype=bf)
    wo[:, :, :hd] = w["o"].reshape(E, H, hd)
    got = O.linear(O.sdpa(qp, kp, vp, scale=hd ** -0.5).permute(0, 2, 1, 3).reshape(1, N, H * hp), wo.reshape(E, H * hp), b["o"])
    assert torch.equal(got, ref)


def test_bunny_config_and_prompt_assembly_follow_the_reference():
    from mlx_vlm_amd.models.llava_bunny import ModelConfig, assemble_input_ids

    cfg = ModelConfig.from_dict(dict(model_type="llava_bunny", auto_map={}, hidden_size=128, mm_hidden_size=144,
                                     num_hidden_layers=2, intermediate_size=256, num_attention_heads=2, rms_norm_eps=1e-6,
                                     vocab_size=1024, vision_config=dict(hidden_size=144)))
    assert cfg.text_config.num_key_value_heads == 2 and cfg.text_config.attention_bias and cfg.text_config.tie_word_embeddings
    assert cfg.vision_config.model_type == "siglip_vision_model" and cfg.image_token_index == -200
    with pytest.raises(ValueError):
        ModelConfig.from_dict(dict(model_type="llava_bunny", auto_map={}, hidden_size=8, mm_hidden_size=8, num_hidden_layers=1,
                                   intermediate_size=8, num_attention_heads=1, rms_norm_eps=1e-6, vocab_size=8,
                                   rope_scaling={"type": "dynamic", "factor": 2.0}, vision_config={}))
    ids, mask = assemble_input_ids(lambda s: [len(w) for w in s.split()], ["aa bbb <image> c", "<image> dddd ee f"], pad_token_id=0)
    assert ids.tolist() == [[2, 3, -200, 1], [-200, 4, 2, 1]] and mask.tolist() == [[1, 1, 1, 1], [1, 1, 1, 1]]


def test_bunny_load_processor_and_prepare_inputs(tmp_path):
    """load_processor for a llava_bunny checkpoint directory: tokenizer with the image processor attached (reference
    utils.py:1260-1270) and the "<image>" branch of prepare_inputs (utils.py:2064-2095)."""
    import json

    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    from mlx_vlm_amd.models.llava_bunny import ModelConfig
    from mlx_vlm_amd.utils import load_config, load_processor, prepare_inputs

    conf = dict(model_type="llava_bunny", auto_map={}, hidden_size=128, mm_hidden_size=144, num_hidden_layers=2,
                intermediate_size=256, num_attention_heads=2, rms_norm_eps=1e-6, vocab_size=64, eos_token_id=[1],
                vision_config=dict(hidden_size=144, num_hidden_layers=2, intermediate_size=288, num_attention_heads=2))
    (tmp_path / "config.json").write_text(json.dumps(conf))
    vocab = {"<unk>": 0, "<eos>": 1, "<pad>": 2, **{f"w{i}": i for i in range(3, 64)}}
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<eos>", pad_token="<pad>", unk_token="<unk>").save_pretrained(str(tmp_path))
    mc = ModelConfig.from_dict(load_config(str(tmp_path)))
    proc = load_processor(str(tmp_path), mc)
    assert proc.tokenizer is proc and proc.image_token_index == -200 and proc.stopping_criteria(1) and not proc.stopping_criteria(5)
    img = np.random.default_rng(0).integers(0, 256, (40, 60, 3), dtype=np.uint8)
    out = prepare_inputs(proc, images=img, prompts="w5 w6 <image> w7")
    assert out["input_ids"].tolist() == [[5, 6, -200, 7]] and out["attention_mask"].tolist() == [[1, 1, 1, 1]]
    assert out["pixel_values"].shape == (1, 3, 384, 384) and out["pixel_values"].dtype == np.float32
    from oracle import llava_bunny as ob
    assert np.array_equal(out["pixel_values"], ob.preprocess([img]))
    text = prepare_inputs(proc, prompts="w9 w10 w11")
    assert text["input_ids"].tolist() == [[9, 10, 11]] and "pixel_values" not in text
    with pytest.raises(ValueError):
        prepare_inputs(proc, images=img, prompts="no placeholder here")


@pytest.mark.parametrize("family", ["qwen2_vl", "llava_bunny"])
def test_model_construction_and_weight_packing_run_without_a_device(family):
    """Config -> model -> synthetic weights -> load_weights on the CPU: arena sizing, q|k|v packing, gate/up interleave,
    head padding / spreading and the KV pool are host logic and must not need a GPU (no kernel is launched).  The decode
    state of the widest batch must fit the small-tensor arena as well."""
    from mlx_vlm_amd import synthetic
    from mlx_vlm_amd.models.qwen2_vl.language import DecodeState

    if family == "qwen2_vl":
        from mlx_vlm_amd.models.qwen2_vl import Model, ModelConfig
        conf = dict(synthetic.QWEN2_VL_2B, hidden_size=256, num_hidden_layers=2, intermediate_size=512, num_attention_heads=2,
                    num_key_value_heads=1, vocab_size=1024,
                    vision_config=dict(synthetic.QWEN2_VL_2B["vision_config"], depth=2, embed_dim=160, hidden_size=256, num_heads=2))
    else:
        from mlx_vlm_amd.models.llava_bunny import Model, ModelConfig
        conf = dict(synthetic.NANOLLAVA, hidden_size=128, mm_hidden_size=144, num_hidden_layers=2, intermediate_size=256,
                    num_attention_heads=2, num_key_value_heads=2, vocab_size=1024,
                    vision_config=dict(synthetic.NANOLLAVA["vision_config"], num_hidden_layers=2, hidden_size=144,
                                       intermediate_size=288, num_attention_heads=2))
    cfg = ModelConfig.from_dict(conf)
    W = synthetic.random_weights(cfg, seed=0, device="cpu")
    model = Model(cfg, device="cpu", kv_pool_tokens=2048, max_seqs=4)
    model.load_weights(W)
    lm = model.language_model
    t = cfg.text_config
    hd = lm.head_dim
    assert hd == 128 and lm.pool.head_dim == 128
    assert lm._w["0.wqkv"].shape == ((t.num_attention_heads + 2 * t.num_key_value_heads) * hd, t.hidden_size)
    assert lm._w["0.wo"].shape == (t.hidden_size, t.num_attention_heads * hd)
    assert lm._w["0.wgu"].shape == (2 * t.intermediate_size, t.hidden_size)
    gate = W["language_model.model.layers.0.mlp.gate_proj.weight"]
    assert torch.equal(lm._w["0.wgu"][0::2], gate) and torch.equal(lm._w["0.wgu"][1], W["language_model.model.layers.0.mlp.up_proj.weight"][0])
    st = DecodeState(lm, 8)
    assert st.h.shape == (8, t.hidden_size) and st.attn.shape == (8, t.num_attention_heads * hd)
    if family == "llava_bunny":
        assert float(lm._w["inv_freq"][32:].abs().max()) == 0.0 and float(lm._w["inv_freq"][0]) == 1.0
        assert model.vision_tower._w["0.wqkv"].shape == (3 * 2 * 80, 144)


def test_generate_step_rejects_unbuilt_options():
    """draft_model / the KV quantisation schemes other than uniform 8-bit group-64 are outside the built path (max_kv_size is
    built since round 4: tests/test_rotating_*.py):
    they raise instead of being dropped silently (reference signature: ar.py:151-214).  Python samplers / logits
    processors and thinking budgets are built since round 4 (the eager step, tests/test_engine_gpu.py); malformed ones raise
    TypeError."""
    import numpy as np
    from mlx_vlm_amd.generate import generate_step

    ids = np.array([[5, 6, 7]])
    for kw in (dict(kv_bits=4), dict(kv_bits=8, kv_group_size=32), dict(kv_bits=3.5), dict(kv_bits=8, kv_quant_scheme="turboquant"),
               dict(draft_model=object())):
        with pytest.raises(NotImplementedError):
            next(generate_step(ids, None, None, None, max_tokens=2, **kw))
    for kw in (dict(logits_processors=[123]), dict(sampler=7)):
        with pytest.raises(TypeError):
            next(generate_step(ids, None, None, None, max_tokens=2, **kw))


def test_dp_batch_generate_carries_model_specific_request_fields(monkeypatch):
    """A request's fields besides input_ids / pixel_values / image_grid_thw / max_tokens (phi3_v: image_sizes, idefics2:
    pixel_attention_mask) travel to the per-rank generator as `extras`, in the order the rank serves its requests."""
    from mlx_vlm_amd import batch, parallel

    seen = {}

    def fake(model, ids, pix, grids, *, max_tokens, stop_ids=(), extras=None, **kw):
        seen.update(ids=ids, pix=pix, grids=grids, max_tokens=max_tokens, extras=extras)
        return [[7] * m for m in max_tokens], None

    monkeypatch.setattr(batch, "generate_batch_continuous", fake)
    reqs = [{"input_ids": np.arange(5), "pixel_values": "pv0", "image_sizes": [[336, 336]], "max_tokens": 2},
            {"input_ids": np.arange(9)},
            {"input_ids": np.arange(7), "pixel_values": "pv2", "pixel_attention_mask": "pm2", "image_grid_thw": "g2"}]
    out = parallel.dp_batch_generate(object(), None, requests=reqs, max_tokens=3)
    order = parallel.shard_requests(3, 0, 1, [5, 9, 7])
    assert [len(t) for t in out["tokens"]] == [2, 3, 3]
    assert seen["extras"] == [{k: v for k, v in reqs[i].items() if k not in parallel._REQUEST_KEYS} for i in order]
    assert seen["pix"] == [reqs[i].get("pixel_values") for i in order] and seen["grids"] == [reqs[i].get("image_grid_thw") for i in order]
    assert seen["extras"][order.index(0)] == {"image_sizes": [[336, 336]]} and seen["extras"][order.index(2)] == {"pixel_attention_mask": "pm2"}


def test_thinking_budget_criteria_matches_reference_known_answers():
    """utils.ThinkingBudgetCriteria against the reference class's own outputs (tests/golden/make_golden_thinking.py executes
    mlx_vlm/utils.py:2252-2335 on seeded token streams): return value, popped forced id and every state field per token."""
    import json

    from mlx_vlm_amd.utils import ThinkingBudgetCriteria

    class Tok:
        def encode(self, t, add_special_tokens=False):
            return {"<think>": [5, 7], "</think>": [8], "\n": [9]}.get(t, [1])

    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "thinking_ref.json")))
    assert len(cases) == 60
    for case in cases:
        c = ThinkingBudgetCriteria(Tok(), **case["kw"])
        for i, (t, pop, want) in enumerate(zip(case["tokens"], case["pops"], case["trace"])):
            ret = c(t)
            popped = c.pop_forced_token_id() if pop else "-"
            assert [ret, popped, c.in_thinking, c.thinking_token_count, c.budget_exceeded, c.forced_token_id] == want, (case["kw"], i)
            if i == 30:
                c.reset_thinking_state()


def test_bench_gpus_2_dry_run_through_the_self_launch():
    """`python bench.py --gpus 2 --dry-run`: the launcher path of the real multi-GPU bench (bench.py re-executes itself under
    torch.distributed.run with 127.0.0.1 rendezvous, refuses a rank count that differs from --gpus), then on 2 gloo CPU
    ranks the weight arena broadcast (contents verified on every rank), dp_batch_generate's deal / gather with a mock
    engine, the timing protocol and ONE JSON line from rank 0.  No kernels run; the scaling run itself is the driver's."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["distributed"]["ranks"] == 2 and out["distributed"]["backend"] == "gloo"
    assert out["distributed"]["weight_broadcast_GBps"] > 0 and out["load"]["weight_bytes"] == 8 * (1 << 21) + 4096 * 4
    assert sorted(out["per_rank_requests"]) == [4, 5] and len(out["serve_s_per_rank"]) == 2 and out["value"] > 0
    assert "launching 2 ranks" in r.stderr
    # a launcher whose rank count disagrees with --gpus is refused (the judge's clock and the line must describe the same job)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_bench_gpus_8_dry_run_has_one_entry_per_rank():
    """`python bench.py --gpus 8 --dry-run`: the driver's 8-rank shape on gloo CPU ranks - eight ranks come up through the
    self-launch, every per-rank vector of the line has eight entries, the requests are dealt over all eight, and every rank
    sized torch's CPU pool for ITS share of the cgroup quota (utils.host_threads_per_rank: quota // LOCAL_WORLD_SIZE)."""
    import json

    from mlx_vlm_amd.utils import cpu_quota

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "64"          # (torchrun would set 1: make the fit do the work)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["distributed"]["ranks"] == 8 and out["config"]["parallelism"] == "dp8"
    for key in ("host_prep_s_per_rank", "decode_s_per_rank", "serve_s_per_rank", "per_rank_requests", "torch_threads_per_rank"):
        assert len(out[key]) == 8, (key, out[key])
    assert sum(out["per_rank_requests"]) == 4 * 8 + 1 and min(out["per_rank_requests"]) >= 4
    assert all(v > 0 for v in out["decode_s_per_rank"])
    share = max(1, cpu_quota() // 8)
    assert all(t <= share for t in out["torch_threads_per_rank"]), (out["torch_threads_per_rank"], share)


def test_host_threads_per_rank_divides_the_quota(monkeypatch):
    from mlx_vlm_amd import utils

    q = utils.cpu_quota()
    for k in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert utils.host_threads_per_rank() == q
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert utils.host_threads_per_rank() == max(1, q // 2)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")          # torchrun's per-node count wins over the job's
    assert utils.host_threads_per_rank() == max(1, q // 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4096")
    assert utils.host_threads_per_rank() == 1


def test_make_sampler_takes_the_whole_reference_surface():
    """make_sampler's arguments (reference sample_utils.py:10-89) all land in the spec; the filters the captured step does not
    carry mark it `extended`, which routes it around an eager step as the sampler callable (generate._resolve_sampler); the
    argument checks of the reference's closures are made at construction."""
    from mlx_vlm_amd.generate import _resolve_sampler
    from mlx_vlm_amd.sample_utils import Sampler, make_sampler

    s = make_sampler(temp=0.7, top_p=0.9, min_p=0.05, top_k=40, seed=3)
    assert not s.extended and s.engine_args() == dict(temperature=0.7, top_p=0.9, min_p=0.05, top_k=40, seed=3)
    assert _resolve_sampler(s, 0.0, 1.0, 0.0, 0, None) == (s, None)
    for kw in (dict(min_tokens_to_keep=4, min_p=0.1), dict(top_n_sigma=1.0), dict(p_less=True), dict(typical_p=0.5),
               dict(xtc_probability=0.3, xtc_threshold=0.1, xtc_special_tokens=[1, 2])):
        e = make_sampler(temp=0.7, seed=1, **kw)
        assert e.extended, kw
        dev, py = _resolve_sampler(e, 0.0, 1.0, 0.0, 0, None)
        assert dev.greedy and py is e
        a = e.sample_args()
        for k, v in kw.items():
            assert a[k] == v, (k, a[k])
        assert not make_sampler(temp=0.0, **kw).extended            # greedy ignores every filter (sample_utils.py:63-64)
    # generate_step's keywords (ar.py:168-170) reach make_sampler
    dev, py = _resolve_sampler(None, 0.8, 0.9, 0.0, 0, 5, top_n_sigma=1.5, p_less=None, typical_p=0.4)
    assert isinstance(py, Sampler) and py.top_n_sigma == 1.5 and py.typical_p == 0.4 and py.top_p == 0.9 and not py.p_less
    for bad in (dict(min_p=1.5), dict(min_tokens_to_keep=0), dict(xtc_probability=0.5, xtc_threshold=0.7), dict(xtc_probability=1.5)):
        with pytest.raises(ValueError):
            make_sampler(temp=0.5, **bad)
    # values make_sampler itself treats as "off" (sample_utils.py:69-74): ignored, as there
    assert not make_sampler(temp=0.5, top_n_sigma=-1.0, typical_p=1.5).extended
