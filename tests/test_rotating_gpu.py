"""`max_kv_size` on MI355X (reference RotatingKVCache, models/cache.py:442-625, built by make_prompt_cache cache.py:45-70 with
keep = 4; generate_step(max_kv_size=), ar.py:173,310-315) through the engine: the window lives in the paged pool as a SET of
tokens - the host keeps the ring (models/cache.py::PagedSequence.rotate_plan, pinned to the reference's held-token sets on the
CPU side) and csrc/kv_rotate.hip moves the entries that have to move.

  * vlm_kv_move_tokens against the pool read back (every layer, every head, K and V layouts);
  * teacher-forced decode through the module contract, every step's logits against the oracle's typed graph of the same
    cache (oracle/qwen2_vl.py max_kv_size=, itself bit-exact against the reference's own files, tests/golden/rotating_ref.npz):
    a prompt shorter than the window (the ring fills during the decode), exactly as long, longer (cut at the first step), and
    a prompt with an image - including the reference's position rule (Qwen2-VL reads the ring's write index as the cache
    offset, language.py:426-431);
  * generate_step(max_kv_size=24) on the peaked head: tokens IDENTICAL to the reference's own generate_step run
    (rotating_ref.npz `gen.*`), log-probs within 2 ulps + 3 % rms;
  * what is not built is refused."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import qwen2_vl as oq
from tests.helpers import bf16_close, build_product_model, synth_request

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotating_ref.npz"))
MAXS = int(G["max_kv_size"])


def _rel_rms(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30))


@pytest.fixture(scope="module")
def tiny():
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    return cfg, W, build_product_model(cfg, W, kv_pool_tokens=8192, max_seqs=16)


def test_kv_move_tokens_moves_every_layer_and_head(tiny):
    import ctypes as C

    from mlx_vlm_amd import _lib

    cfg, W, model = tiny
    lm = model.language_model
    pool = lm.pool
    cache = lm.make_cache()
    seq = cache[0]._seq
    ids = np.random.default_rng(3).integers(3, 1000, (1, 150))
    lm(ids, cache=cache, logits_to_keep=1)
    before = [tuple(t.clone() for t in cache[i].state) for i in range(len(cache))]
    src = np.array([149, 70, 5, 64], dtype=np.int32)
    dst = np.array([1, 130, 63, 128], dtype=np.int32)            # crosses pages both ways; disjoint from src
    dev = _lib.h2d(np.stack([np.full(4, seq.seq, dtype=np.int32), src, dst]), lm.device)
    _lib.check(_lib.lib().vlm_kv_move_tokens(pool.kpool.data_ptr(), pool.vpool.data_ptr(), pool.layer_stride, pool.n_layers,
                                             dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), 4, pool.block_table.data_ptr(),
                                             pool.max_pages, pool.n_kv_heads, pool.head_dim,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "kv_move_tokens")
    torch.cuda.synchronize()
    for i in range(len(cache)):
        k, v = cache[i].state
        ek, ev = before[i][0].clone(), before[i][1].clone()
        for a, d in zip(src, dst):
            ek[:, :, d], ev[:, :, d] = before[i][0][:, :, a], before[i][1][:, :, a]
        assert torch.equal(k, ek) and torch.equal(v, ev), i
    seq.release()


def _engine_teacher_forced(model, ids, pix, thw, forced, max_kv_size):
    from mlx_vlm_amd.models import cache as cache_mod

    lm = model.language_model
    kw = dict(image_grid_thw=thw) if thw is not None else {}
    f = model.get_input_embeddings(ids, torch.from_numpy(pix) if pix is not None else None, **kw)
    cache = cache_mod.make_prompt_cache(lm, max_kv_size=max_kv_size)
    out = lm(ids, f.inputs_embeds, cache=cache, position_ids=f.position_ids, rope_deltas=f.rope_deltas, logits_to_keep=1)
    rows = [out.logits[0, -1].clone()]
    seq = cache[0]._seq
    for y in forced:
        rows.append(lm(np.array([[int(y)]]), cache=cache).logits[0, -1].clone())
        assert seq.kv_entries <= max(max_kv_size, 1) and cache[0].size() == min(seq.offset, max_kv_size)
    n, pages = seq.offset, len(seq.pages)
    seq.release()
    return torch.stack(rows), n, pages


@pytest.mark.parametrize("name", ["short", "exact", "long"])
def test_teacher_forced_decode_over_the_window_every_step(tiny, name):
    """The golden prompts (9 / 24 / 41 tokens at max_kv_size 24), 30 forced tokens: every row within 2e-2 rel-rms of the oracle
    over the SAME window, and - once tokens have left it - farther from the unbounded-cache oracle than from that one."""
    cfg, W, model = tiny
    ids, forced = G[f"tf.{name}.input_ids"], G["tf.forced"]
    got, n, pages = _engine_teacher_forced(model, ids, None, None, forced, MAXS)
    assert n == ids.shape[1] + len(forced) and pages <= 2
    ref = oq.decode_teacher_forced(W, cfg, ids, None, None, forced, max_kv_size=MAXS)
    errs = [_rel_rms(got[i], ref[i]) for i in range(ref.shape[0])]
    assert max(errs) < 2e-2, (name, max(errs), errs)
    # against the reference's own rows (pure-MLX rotation; the fused form is its contract-level equal): same bound + the
    # difference of the two rotations
    assert max(_rel_rms(got[i], torch.from_numpy(G[f"tf.{name}.logits"][i])) for i in range(ref.shape[0])) < 4e-2
    plain = oq.decode_teacher_forced(W, cfg, ids, None, None, forced)
    parted = [i for i, same in enumerate(G[f"tf.{name}.equals_unbounded"]) if not same]
    tail = parted[2:]
    d_w = np.mean([_rel_rms(got[i], ref[i]) for i in tail])
    d_p = np.mean([_rel_rms(got[i], plain[i]) for i in tail])
    assert d_w < 0.5 * d_p, (name, d_w, d_p)
    print(f"rotating window {name}: worst row rel-rms {max(errs):.4f}; past the wrap: {d_w:.4f} to the windowed oracle, {d_p:.4f} to the unbounded one")


def test_teacher_forced_decode_over_the_window_with_an_image(tiny):
    """an image prompt (rope delta != 0) longer than the window, 70 forced tokens (the ring wraps three times)"""
    cfg, W, model = tiny
    ids, pix, thw = synth_request(cfg, [(56, 84)], n_text=14, seed=44)
    forced = np.random.default_rng(45).integers(3, 1000, 70)
    maxs = 20
    assert ids.shape[1] > maxs
    got, n, pages = _engine_teacher_forced(model, ids, pix, thw, forced, maxs)
    ref = oq.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pix).to(BF), thw, forced, max_kv_size=maxs)
    errs = [_rel_rms(got[i], ref[i]) for i in range(ref.shape[0])]
    assert max(errs) < 2e-2, (max(errs), errs)
    plain = oq.decode_teacher_forced(W, cfg, ids, torch.from_numpy(pix).to(BF), thw, forced)
    assert np.mean([_rel_rms(got[i], ref[i]) for i in range(3, 71)]) < 0.5 * np.mean([_rel_rms(got[i], plain[i]) for i in range(3, 71)])


def _peaked_tiny():
    cfg = oq.tiny_cfg()
    cfg.text.tie_word_embeddings = False
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    for k in list(W):
        if k.endswith("o_proj.weight") or k.endswith("down_proj.weight"):
            W[k] = (W[k].float() * 0.5).to(BF)
    return cfg, oq.peak_head(W, cfg, gamma=4.0, stride=389, n_cycle=1000)


@pytest.mark.parametrize("name", ["short", "long"])
def test_generate_step_with_max_kv_size_gives_the_references_tokens(name):
    from mlx_vlm_amd.generate import generate_step

    cfg, W = _peaked_tiny()
    model = build_product_model(cfg, W, kv_pool_tokens=4096, max_seqs=4)
    ids = G[f"gen.{name}.input_ids"]
    ref_toks, ref_lp = G[f"gen.{name}.tokens"].tolist(), torch.from_numpy(G[f"gen.{name}.logprobs"])
    toks, lps = [], []
    for t, lp in generate_step(ids, model, None, None, max_tokens=len(ref_toks), temperature=0.0, max_kv_size=MAXS):
        toks.append(t)
        lps.append(lp.float().cpu())
    assert toks == ref_toks                                   # the reference's own generate_step(max_kv_size=24) run
    for i in range(len(toks)):
        ok, rep = bf16_close(lps[i], ref_lp[i].to(BF), ulps=2, atol_rms=3e-2)
        assert ok, (i, rep)
    # and the bound does change the function: the unbounded run's log-probs part from these once the window has dropped tokens
    plain = [lp.float().cpu() for _, lp in generate_step(ids, model, None, None, max_tokens=len(ref_toks), temperature=0.0)]
    assert any(not torch.equal(a, b) for a, b in zip(lps[-8:], plain[-8:]))


def test_what_the_window_does_not_cover_is_refused(tiny):
    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.generate import generate_step
    from mlx_vlm_amd.models import cache as cache_mod

    cfg, W, model = tiny
    ids = np.random.default_rng(9).integers(3, 1000, (1, 40))
    with pytest.raises(NotImplementedError):                  # the reference: "RotatingKVCache Quantization NYI"
        next(generate_step(ids, model, None, None, max_tokens=4, max_kv_size=16, kv_bits=8, quantized_kv_start=0))
    with pytest.raises(NotImplementedError):                  # chunked prefill over a rotating cache
        next(generate_step(ids, model, None, None, max_tokens=4, max_kv_size=16, prefill_step_size=32))
    with pytest.raises(ValueError):                           # the reference's batch path refuses keep > 0 (ar.py:831-834)
        BatchGenerator(model, None, max_kv_size=16)
    lm = model.language_model
    cache = cache_mod.make_prompt_cache(lm, max_kv_size=16)
    lm(ids, cache=cache, logits_to_keep=1)
    with pytest.raises(NotImplementedError):                  # a second multi-token update of the window
        lm(ids[:, :5], cache=cache, logits_to_keep=1, position_ids=np.broadcast_to(np.arange(40, 45), (3, 1, 5)))
    cache[0]._seq.release()


def test_max_kv_size_bounds_the_kv_reservation():
    """the prefill of a rotating sequence reserved prompt + max_tokens + 2 tokens of pages, so a long generation
    ran out of pages where the reference runs in max_size entries.  A pool whose sequences hold at most 2 pages (128 tokens):
    a 40-token prompt decodes 300 tokens under max_kv_size=24, and the sequence never owns more than one page."""
    from mlx_vlm_amd.generate import generate_step
    from mlx_vlm_amd.models import cache as cache_mod

    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    model = build_product_model(cfg, W, kv_pool_tokens=1024, max_seqs=4)
    lm = model.language_model
    lm.pool.max_pages = 2                               # (the block table is wider; the allocator refuses a third page)
    ids = np.random.default_rng(11).integers(3, 1000, (1, 40))
    cache = cache_mod.make_prompt_cache(lm, max_kv_size=24)
    toks = [t for t, _ in generate_step(ids, model, None, None, max_tokens=300, temperature=0.0, prompt_cache=cache)]
    assert len(toks) == 300 and len(cache[0]._seq.pages) == 1 and cache[0].offset == 40 + 299
    with pytest.raises(RuntimeError, match="max_pages_per_seq"):          # the unbounded cache does need the pages
        list(generate_step(ids, model, None, None, max_tokens=300, temperature=0.0))


def test_phi3v_refuses_a_rotating_prompt_longer_than_the_window():
    """the reference's Phi-3.5-V passes cache[0] to create_attention_mask (phi3_v.py:163), so a first prompt longer
    than max_kv_size is prefilled under RotatingKVCache.make_mask's WINDOWED causal mask (cache.py:586-596); this engine's prefill
    is plain causal - the case is refused, the prompt that fits the window runs."""
    from mlx_vlm_amd.models.phi3_v.language import LanguageModel as PhiLM
    from mlx_vlm_amd.models.qwen2_vl.language import LanguageModel as QwenLM

    assert PhiLM.ROTATING_PROMPT_WINDOW_MASK is True and QwenLM.ROTATING_PROMPT_WINDOW_MASK is False
    from oracle import phi3_v as op
    from tests.helpers import build_phi3v_model
    from tests.test_vlm_family_phi3v_gpu import SCALES
    from mlx_vlm_amd.models import cache as cache_mod

    cfg = op.tiny_cfg()
    model = build_phi3v_model(cfg, op.random_weights(cfg, seed=4321, dtype=BF, **SCALES), kv_pool_tokens=4096, max_seqs=4)
    lm = model.language_model
    ids = np.random.default_rng(5).integers(3, 200, (1, 40))
    cache = cache_mod.make_prompt_cache(lm, max_kv_size=16)
    with pytest.raises(NotImplementedError, match="sliding-window mask"):
        lm(ids, cache=cache, logits_to_keep=1)
    cache[0]._seq.release()
    cache = cache_mod.make_prompt_cache(lm, max_kv_size=64)
    out = lm(ids, cache=cache, logits_to_keep=1)
    assert out.logits.shape[1] == 1 and cache[0].offset == 40
    cache[0]._seq.release()
