"""The whole make_sampler surface (reference mlx_vlm/sample_utils.py:10-89) on MI355X through vlm_sample_ex: top-n-sigma,
p-less, locally typical, top-p, min-p with min_tokens_to_keep, XTC, top-k - the FILTERED log-probs the kernel leaves in its
scratch row against

  * the reference's own functions' outputs (tests/golden/samplers_ref.npz: its sample_utils.py executed over the shim) on
    the golden bf16 rows, bit for bit;
  * the oracle's typed restatement (oracle/ops.py::sampler_filters, pinned to the same vectors on the CPU side) on rows of
    real vocabulary sizes (32,003 ragged / 151,936) and on chains of filters in make_sampler's order.

Agreement is EXACT (which tokens survive, and their values) except where a float32 sum taken in another order lands on the
other side of a bf16 rounding edge: a survivor set may differ by elements whose deciding quantity sits on the filter's
threshold - the tests name that quantity per filter and accept nothing else.  The draw (Gumbel-max over the filtered row
with the counter hash) is checked against oracle.categorical_gumbel on the kernel's own filtered row."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "samplers_ref.npz"))


@pytest.fixture(scope="module")
def vops():
    from mlx_vlm_amd import ops
    return ops


def _rows(B, V, seed, scales=(1.5, 5.0, 2.0)):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, V, generator=g) * torch.tensor([[scales[b % len(scales)]] for b in range(B)])
    if B > 2:
        x[2] = torch.round(x[2] * 2) / 2            # exact ties
    return (x - torch.logsumexp(x, -1, keepdim=True)).to(BF)


def _filtered(vops, lp, temp=0.8, seed=11, step=0, **kw):
    st = torch.tensor([step], dtype=torch.int32, device="cuda")
    tok, _, filt = vops.sample(lp.cuda(), temperature=temp, seed=seed, step=st, want_logprobs=False, input_is_logprobs=True,
                               return_filtered=True, **kw)
    torch.cuda.synchronize()
    return tok.cpu(), filt.cpu()


def _kept(x):
    return torch.isfinite(x.float())


def _assert_same(got, ref, lp, what=""):
    """got / ref: filtered rows [B, V].  Survivors keep their values; the survivor sets are equal, or differ only in EDGE
    elements: an element the two sides disagree on has, within one bf16 step of its own log-prob, an element the reference puts
    in the other class (every filter here cuts in log-prob order, or - typical-p - in a band of it whose two ends are such
    edges).  -> number of elements on the other side of an edge."""
    kg, kr = _kept(got), _kept(ref)
    assert torch.equal(got.float()[kg], lp.float()[kg]), what          # a filter never changes a survivor
    if torch.equal(kg, kr):
        return 0
    q = lp.float()
    n_bad = 0
    for b in range(lp.shape[0]):
        diff = torch.nonzero(kg[b] != kr[b]).flatten()
        if diff.numel() == 0:
            continue
        assert diff.numel() <= max(4, int(0.01 * int(kr[b].sum()))), (what, b, int(diff.numel()), int(kg[b].sum()), int(kr[b].sum()))
        for i in diff.tolist():
            v = float(q[b, i])
            other = q[b, ~kr[b] & torch.isfinite(q[b])] if bool(kr[b, i]) else q[b, kr[b]]
            d = float((other - v).abs().min()) if other.numel() else float("inf")
            assert d <= 2.0 ** -7 * abs(v) + 1e-30, (what, b, i, v, d, int(kg[b].sum()), int(kr[b].sum()))
        n_bad += int(diff.numel())
    return n_bad


def test_filters_equal_the_reference_on_its_golden_rows(vops):
    """Kernel output vs the reference's own apply_* outputs (bf16 inputs), every golden case."""
    lp = torch.from_numpy(G["bf16.logprobs"]).to(BF)
    sp = [int(v) for v in G["xtc_special"]]
    cases = []
    for ns in (0.5, 1.5):
        cases.append((f"top_n_sigma_{ns}", dict(top_n_sigma=ns), 0.8))
    for temp in (0.7, 1.3):
        cases.append((f"p_less_{temp}", dict(p_less=True), temp))
    for tp in (0.3, 0.9):
        cases.append((f"typical_p_{tp}", dict(typical_p=tp), 0.8))
    for mp, keep in ((0.3, 4), (0.05, 1), (0.9, 7)):
        cases.append((f"min_p_{mp}_keep_{keep}", dict(min_p=mp, min_tokens_to_keep=keep), 0.8))
    for tp in (0.5, 0.9, 0.99):
        cases.append((f"top_p_{tp}", dict(top_p=tp), 0.8))
    cases.append(("top_k_5", dict(top_k=5), 0.8))
    total = 0
    for name, kw, temp in cases:
        ref = torch.from_numpy(G[f"bf16.{name}"])
        _, got = _filtered(vops, lp, temp=temp, **kw)
        n = int((_kept(got) != _kept(ref)).sum())
        total += n
        assert n == 0, (name, n, [int(_kept(got[b]).sum()) for b in range(3)], [int(_kept(ref[b]).sum()) for b in range(3)])
        assert torch.equal(got.float()[_kept(ref)], ref[_kept(ref)]), name
    # xtc: one row per call; probability 1.0 = always (any draw), 0.0 = the filter is off (make_sampler adds it only when > 0)
    for thr in (0.02, 0.08):
        ref = torch.from_numpy(G[f"bf16.xtc_{thr}"])
        for r in range(3):
            _, got = _filtered(vops, lp[r:r + 1], xtc_probability=1.0, xtc_threshold=thr, xtc_special_tokens=sp)
            assert torch.equal(_kept(got[0]), _kept(ref[r])), (thr, r, int(_kept(got[0]).sum()), int(_kept(ref[r]).sum()))
            assert all(bool(_kept(got[0])[t]) for t in sp)
    # the CHAINS: what the reference's make_sampler closure hands to its draw (filters in ITS order)
    import json
    chains = json.loads(str(G["chains_json"]))
    for ci, kw in enumerate(chains):
        ref = torch.from_numpy(G[f"bf16.chain_{ci}"])
        if kw.get("xtc_probability"):
            got = torch.cat([_filtered(vops, lp[r:r + 1], **kw)[1] for r in range(3)])
        else:
            _, got = _filtered(vops, lp, **kw)
        assert torch.equal(_kept(got), _kept(ref)), (ci, kw, [int(_kept(got[b]).sum()) for b in range(3)], [int(_kept(ref[b]).sum()) for b in range(3)])
    print(f"golden rows: {len(cases)} filter cases + 6 xtc calls + {len(chains)} make_sampler chains, survivor sets identical")


@pytest.mark.parametrize("V", [32003, 151936])
@pytest.mark.parametrize("name,kw,temp", [
    ("top_n_sigma", dict(top_n_sigma=1.0), 0.8),
    ("p_less", dict(p_less=True), 0.7),
    ("p_less_hot", dict(p_less=True), 1.6),
    ("typical_p", dict(typical_p=0.5), 0.8),
    ("typical_p_wide", dict(typical_p=0.95), 0.8),
    ("top_p", dict(top_p=0.9), 0.8),
    ("top_p_low", dict(top_p=0.25), 0.8),
    ("min_p", dict(min_p=0.05), 0.8),
    ("min_p_keep", dict(min_p=0.6, min_tokens_to_keep=300), 0.8),
    ("top_k", dict(top_k=50), 0.8),
    ("chain_classic", dict(top_p=0.95, min_p=0.02, top_k=64), 1.0),
    ("chain_typical", dict(typical_p=0.9, top_p=0.9, top_k=40), 0.9),
    ("chain_sigma", dict(top_n_sigma=2.0, min_p=0.1, min_tokens_to_keep=3, top_k=16), 1.1),
])
def test_filters_equal_the_oracle_at_vocabulary_size(vops, V, name, kw, temp):
    lp = _rows(3, V, seed=200 + len(name))
    _, got = _filtered(vops, lp, temp=temp, **kw)
    ref = O.sampler_filters(lp, temp, **kw)
    n = _assert_same(got, ref, lp, what=name)
    print(f"{name} V={V}: kept {[int(_kept(ref[b]).sum()) for b in range(3)]}, elements on the other side of an edge: {n}")


@pytest.mark.parametrize("kw", [dict(top_p=0.9), dict(top_k=33), dict(min_p=0.5, min_tokens_to_keep=200),
                                dict(top_p=0.95, min_p=0.01, top_k=100)])
def test_filters_on_rows_with_positive_values(vops, kw):
    """A sampler closure takes whatever row it is handed (the reference's make_sampler documents top-n-sigma / p-less on raw
    logits): rows with positive entries - keys the LDS half of the histogram does not hold - through the same filters."""
    g = torch.Generator().manual_seed(321)
    x = (torch.randn(2, 40000, generator=g) * 3.0 + 1.0).to(BF)       # unnormalised: about 60 % positive
    x[1] = (x[1].float() - 10.5).to(BF)                               # second row: only a handful above zero
    _, got = _filtered(vops, x, **kw)
    ref = O.sampler_filters(x, 0.8, **kw)
    n = _assert_same(got, ref, x, what=str(kw))
    print(f"{kw}: kept {[int(_kept(ref[b]).sum()) for b in range(2)]}, on the other side of an edge: {n}")


def test_xtc_draw_and_special_tokens(vops):
    """xtc_probability = 0.5 over 24 steps: applied exactly on the steps where the counter hash says so (oracle.xtc_draw),
    removing everything above the weakest above-threshold token except the special ones."""
    V = 32003
    lp = _rows(1, V, seed=77, scales=(4.0,))
    sp = [int(i) for i in torch.topk(lp[0].float(), 3).indices[:2]] + [5]
    applied = 0
    for step in range(24):
        _, got = _filtered(vops, lp, seed=1234, step=step, xtc_probability=0.5, xtc_threshold=0.01, xtc_special_tokens=sp)
        ref = O.sampler_filters(lp, 0.8, xtc_probability=0.5, xtc_threshold=0.01, xtc_special_tokens=sp, seed=1234, step=step)
        _assert_same(got, ref, lp, what=f"xtc step {step}")
        on = not (np.float32(O.xtc_draw(1234, step)) > np.float32(0.5))
        assert (int(_kept(got).sum()) < V) == on, step
        applied += on
    assert 4 < applied < 20
    with pytest.raises(Exception):       # the reference's minimum runs over the whole array: one row per call
        _filtered(vops, _rows(2, 1024, seed=3), xtc_probability=1.0, xtc_threshold=0.05)


def test_typical_p_after_a_filter_follows_the_reference_into_its_nan_order(vops):
    """The reference's typical-p on a row that already holds -inf: p * logp = 0 * -inf = NaN, the entropy is NaN, every sort
    key is NaN, the (stable) order is the index order and tokens are kept in INDEX order until the mass before them reaches
    typical_p.  Not a useful filter - but it is what make_sampler(top_n_sigma=, typical_p=) computes, and IEEE arithmetic in
    the kernel arrives at the same survivors (up to the float32 order of the running sum at the cut)."""
    lp = _rows(2, 32003, seed=41)
    _, got = _filtered(vops, lp, top_n_sigma=1.5, typical_p=0.9)
    ref = O.sampler_filters(lp, 0.8, top_n_sigma=1.5, typical_p=0.9)
    for b in range(2):
        kg, kr = _kept(got[b]), _kept(ref[b])
        assert int((kg != kr).sum()) <= 2, (b, int(kg.sum()), int(kr.sum()))
        assert 0 < int(kr.sum()) < 2000


def test_the_draw_runs_over_the_filtered_row(vops):
    """tokens = Gumbel-max over (filtered log-probs / temp) with the counter hash at (seed, step, row, index)."""
    V = 32003
    lp = _rows(2, V, seed=91)
    agree = 0
    for step in range(30):
        tok, filt = _filtered(vops, lp, temp=0.9, seed=5, step=step, typical_p=0.9, top_k=50)
        for b in range(2):
            assert bool(_kept(filt[b])[int(tok[b])])
            agree += int(int(tok[b]) == O.categorical_gumbel(filt[b], 0.9, seed=5, step=step, row=b))
    assert agree >= 58, agree


def test_sampler_spec_is_callable_on_logprobs_and_generate_step_takes_the_keywords(vops):
    """make_sampler(...) with the extra filters: callable on a log-probs tensor (the reference's closure contract), and
    generate_step(top_n_sigma= / p_less= / typical_p=) routes to it (ar.py:168-170,279-288) - every token drawn is a
    survivor of the oracle's filter chain on that step's log-probs."""
    from mlx_vlm_amd.generate import generate_step
    from mlx_vlm_amd.sample_utils import make_sampler
    from oracle import qwen2_vl as oq
    from tests.helpers import build_product_model

    smp = make_sampler(temp=0.8, typical_p=0.6, min_p=0.05, min_tokens_to_keep=3, seed=9)
    assert smp.extended
    lp = _rows(2, 4096, seed=17)
    ref = O.sampler_filters(lp, 0.8, typical_p=0.6, min_p=0.05, min_tokens_to_keep=3)
    for _ in range(20):
        tok = smp(lp.cuda()).cpu()
        assert all(bool(_kept(ref[b])[int(tok[b])]) for b in range(2))
    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    model = build_product_model(cfg, W, kv_pool_tokens=4096, max_seqs=8)
    ids = np.random.default_rng(5).integers(3, 1000, (1, 21))
    out = list(generate_step(ids, model, None, None, max_tokens=12, temperature=0.9, typical_p=0.8, top_k=30, seed=3))
    assert len(out) == 12
    for t, lpv in out:
        row = lpv.reshape(1, -1).cpu()
        keep = _kept(O.sampler_filters(row, 0.9, typical_p=0.8, top_k=30))[0]
        near = row.float()[0, t] >= row.float()[0][keep].min() - 2.0 ** -6 * abs(float(row.float()[0][keep].min()))
        assert bool(keep[t]) or bool(near), t


def test_batch_generator_runs_a_sampler_with_the_extra_filters(vops):
    """BatchGenerator(sampler=make_sampler(top_n_sigma=..., min_tokens_to_keep=...)) (reference ar.py:2584-2606 takes any
    sampler): the filters the captured step does not carry run through vlm_sample_ex around an eager step, all rows in one
    call; every token drawn is a survivor of the oracle's chain on that row's log-probs (or sits on its edge)."""
    import dataclasses

    from mlx_vlm_amd.batch import BatchGenerator
    from mlx_vlm_amd.sample_utils import Sampler, make_sampler
    from oracle import qwen2_vl as oq
    from tests.helpers import build_product_model

    seen = []

    class Recording(Sampler):
        def __call__(self, logprobs):
            tok = super().__call__(logprobs)
            seen.append((logprobs.detach().to(BF).cpu(), tok.cpu()))
            return tok

    cfg = oq.tiny_cfg()
    W = oq.random_weights(cfg, seed=1234, dtype=BF, std=0.05, embed_std=0.2)
    model = build_product_model(cfg, W, kv_pool_tokens=8192, max_seqs=24)
    kw = dict(top_n_sigma=1.5, min_p=0.3, min_tokens_to_keep=5)
    smp = Recording(**dataclasses.asdict(make_sampler(temp=0.9, seed=21, **kw)))
    assert smp.extended
    gen = BatchGenerator(model, None, max_tokens=6, completion_batch_size=4, prefill_batch_size=2, sampler=smp)
    rng = np.random.default_rng(8)
    uids = gen.insert([rng.integers(3, 1000, n) for n in (9, 17, 30, 12, 25)], [6, 5, 6, 4, 6])
    got = {u: [] for u in uids}
    while gen.has_work:
        _, out = gen.next()
        for r in out:
            got[r.uid].append(r.token)
    gen.close()
    assert [len(got[u]) for u in uids] == [6, 5, 6, 4, 6]
    assert len(seen) >= 5
    n = 0
    for lp, tok in seen:
        lp2 = lp.reshape(-1, lp.shape[-1])
        ref = O.sampler_filters(lp2, 0.9, **kw)
        for b in range(lp2.shape[0]):
            keep = _kept(ref[b])
            lo = float(lp2[b].float()[keep].min())
            t = int(tok.reshape(-1)[b])
            assert bool(keep[t]) or float(lp2[b, t]) >= lo - 2.0 ** -6 * abs(lo), (b, t)
            n += 1
    assert n >= 20


_SPLIT_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from mlx_vlm_amd import ops
from tests.test_sampler_gpu import _split_cases, _filtered, _split_logit_cases, _from_logits
out = {}
for name, lp, kw in _split_cases():
    tok, filt = _filtered(ops, lp, **kw)
    out[name] = filt.view(torch.int16).numpy(); out[name + ".tok"] = tok.numpy()
for name, logits, kw in _split_logit_cases():
    tok, lp, filt = _from_logits(ops, logits, **kw)
    out[name] = filt.view(torch.int16).numpy(); out[name + ".tok"] = tok.numpy(); out[name + ".lp"] = lp.view(torch.int16).numpy()
np.savez(sys.argv[2], **out)
"""


def _split_logit_cases():
    """LOGITS in (the captured decode step's form): the log-prob pass rides in the histogram launch and the draw in the mask
    launch.  A 3-row call at the real vocabulary, a row with -inf logits, an odd leading dimension handled by the fallback."""
    g = torch.Generator().manual_seed(77)
    big = (torch.randn(3, 151936, generator=g) * 2.5 + 3.0).to(BF)
    holes = (torch.randn(2, 32768, generator=g) * 2.0).to(BF)
    holes[:, 5::7] = float("-inf")
    return [("logits_v151936_p0.9", big, dict(top_p=0.9)), ("logits_v151936_p0.3", big, dict(top_p=0.3, temp=1.3, seed=5, step=9)),
            ("logits_holes", holes, dict(top_p=0.8)), ("logits_chain", big, dict(top_p=0.9, min_p=0.02, top_k=50)),
            ("logits_holes_chain", holes, dict(min_p=0.05, top_k=11))]


def _from_logits(vops, logits, temp=0.8, seed=11, step=0, **kw):
    st = torch.tensor([step], dtype=torch.int32, device="cuda")
    tok, lp, filt = vops.sample(logits.cuda(), temperature=temp, seed=seed, step=st, want_logprobs=True, return_filtered=True, **kw)
    torch.cuda.synchronize()
    return tok.cpu(), lp.cpu(), filt.cpu()


def _split_cases():
    """rows for the split path (csrc/sample.hip: top-p / min-p / top-k, V % 8 == 0, V >= 8192): random rows at three temperatures of
    the distribution, a row of exact ties (hundreds of elements share the crossing key: the rank inside the bin decides), a row
    with removed (-inf) tokens, rows with positive values (the global half of the histogram), a top_p so small that nothing
    crosses, a 3-row call (one launch, three rows)"""
    cases = []
    lp = _rows(3, 151936, seed=901)
    for p in (0.9, 0.5, 0.999, 0.05):
        cases.append((f"v151936_p{p}", lp, dict(top_p=p)))
    ties = torch.round(torch.randn(2, 65536, generator=torch.Generator().manual_seed(5)) * 1.5) * 0.25
    cases.append(("ties", (ties - torch.logsumexp(ties, -1, keepdim=True)).to(BF), dict(top_p=0.8)))
    holes = _rows(2, 32768, seed=77).clone()
    holes[:, ::3] = float("-inf")
    cases.append(("holes", holes, dict(top_p=0.7)))
    g = torch.Generator().manual_seed(321)
    pos = (torch.randn(2, 40000, generator=g) * 3.0 + 1.0).to(BF)
    pos[1] = (pos[1].float() - 10.5).to(BF)
    cases.append(("positive", pos, dict(top_p=0.9)))
    cases.append(("no_crossing", _rows(1, 16384, seed=3), dict(top_p=0.001)))
    # round 6: the chain top-p -> min-p -> top-k folded into the crossing launch (any subset, min_tokens_to_keep = 1)
    cases.append(("chain", lp, dict(top_p=0.9, min_p=0.02, top_k=50)))
    cases.append(("chain_small_k", lp, dict(top_p=0.95, min_p=0.001, top_k=3)))
    cases.append(("min_p", lp, dict(min_p=0.05)))
    cases.append(("top_k", lp, dict(top_k=40)))
    cases.append(("min_p_top_k", lp, dict(min_p=0.01, top_k=1000)))
    cases.append(("top_p_top_k", lp, dict(top_p=0.5, top_k=7)))
    tl = (ties - torch.logsumexp(ties, -1, keepdim=True)).to(BF)
    for k in (1, 100, 1000, 5000, 30000):                  # the k-th value inside bins of hundreds: top-p's bin, another, none
        cases.append((f"ties_chain_k{k}", tl, dict(top_p=0.8, min_p=0.1, top_k=k)))
        cases.append((f"ties_top_k{k}", tl, dict(top_k=k)))
    cases.append(("ties_min_p", tl, dict(min_p=0.3, top_p=0.99)))
    cases.append(("holes_k_beyond", holes, dict(top_k=30000)))      # more than the finite tokens: nothing to remove
    cases.append(("holes_chain", holes, dict(top_p=0.7, min_p=0.05, top_k=20)))
    cases.append(("positive_chain", pos, dict(top_p=0.9, min_p=0.02, top_k=64)))
    cases.append(("positive_min_p", pos, dict(min_p=0.1)))
    cases.append(("no_crossing_chain", _rows(1, 16384, seed=3), dict(top_p=0.001, min_p=0.05, top_k=9)))
    return cases


def test_split_top_p_equals_the_one_workgroup_kernel(vops, tmp_path):
    """The row-split top-p path (four short launches over 64 workgroups) against the one-workgroup kernel it replaces
    (VLM_SAMPLE_SPLIT=0, run in a child process: the knob is read once): filtered rows AND drawn tokens bit for bit; then the
    split path again on a REUSED workspace with a top-k call (the one-workgroup kernel, which touches the global histogram when
    a row holds positive values) between two top-p calls - the histogram must be all zero between calls."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_path = str(tmp_path / "single.npz")
    env = dict(os.environ, VLM_SAMPLE_SPLIT="0")
    r = subprocess.run([sys.executable, "-c", _SPLIT_CHILD, root, ref_path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(ref_path)
    for name, lp, kw in _split_cases():
        tok, filt = _filtered(vops, lp, **kw)
        assert np.array_equal(filt.view(torch.int16).numpy(), ref[name]), (name, int((filt.view(torch.int16).numpy() != ref[name]).sum()))
        assert np.array_equal(tok.numpy(), ref[name + ".tok"]), name
    for name, logits, kw in _split_logit_cases():
        tok, lp, filt = _from_logits(vops, logits, **kw)
        assert np.array_equal(lp.view(torch.int16).numpy(), ref[name + ".lp"]), name          # the log-prob pass inside launch A
        assert np.array_equal(filt.view(torch.int16).numpy(), ref[name]), (name, int((filt.view(torch.int16).numpy() != ref[name]).sum()))
        assert np.array_equal(tok.numpy(), ref[name + ".tok"]), name                           # the draw inside launch D
    # reused workspace, interleaved with the other kernel on rows with positive values
    pos = [lp for name, lp, _ in _split_cases() if name == "positive"][0].cuda()
    ws = vops.sample_workspace(2, "cuda")
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    outs = []
    for kw in (dict(top_p=0.9), dict(top_k=33, min_p=0.01, min_tokens_to_keep=2), dict(top_p=0.9), dict(top_p=0.9)):
        _, _, filt = vops.sample(pos, temperature=0.8, seed=11, step=st, want_logprobs=False, input_is_logprobs=True,
                                 return_filtered=True, ws=ws, **kw)
        outs.append(filt.clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16)) and torch.equal(outs[2].view(torch.int16), outs[3].view(torch.int16))
    assert np.array_equal(outs[0].cpu().view(torch.int16).numpy(), ref["positive"])
    row_w, row_hist = 4 * 64 + 65536 + 128, 4 * 64       # csrc/sample.hip ROW_W / ROW_HIST: one block of 4-byte words per row
    for b in range(2):
        hist = ws[256 + 4 * (b * row_w + row_hist): 256 + 4 * (b * row_w + row_hist + 65536)].view(torch.int32)
        assert int(hist.abs().sum()) == 0                  # all zero between calls


def test_split_top_p_on_a_workspace_stepped_at_changing_widths(vops):
    """batch.py steps ONE DecodeState.sample_ws at varying widths.  The workspace is laid out in per-row blocks at
    fixed offsets (csrc/sample.hip ROW_W), so a step over the first 4 rows of a workspace that just served 8 finds its
    histograms zero and its control words re-armed: top-p at B = 8, then 4, then 8 on one workspace gives the rows and tokens of
    fresh workspaces, and between the calls a top-k step (the one-workgroup kernel) runs at yet another width."""
    lp8 = _rows(8, 32768, seed=21).cuda()
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    kw = dict(temperature=0.8, seed=5, step=st, want_logprobs=False, input_is_logprobs=True, return_filtered=True, top_p=0.9)

    def fresh(rows):
        tok, _, filt = vops.sample(lp8[:rows], ws=vops.sample_workspace(rows, "cuda"), **kw)
        return tok.clone(), filt.clone()

    want8, want4 = fresh(8), fresh(4)
    ws = vops.sample_workspace(8, "cuda")
    for rows, want in ((8, want8), (4, want4), (8, want8), (4, want4)):
        tok, _, filt = vops.sample(lp8[:rows], ws=ws, **kw)
        assert torch.equal(tok, want[0]), rows
        assert torch.equal(filt.view(torch.int16), want[1].view(torch.int16)), rows
        # another filter combination at another width in between (sample_filter_kernel: its own use of the histogram words)
        vops.sample(lp8[:6], ws=ws, **dict(kw, top_p=1.0, top_k=17, min_p=0.01, min_tokens_to_keep=2))
    torch.cuda.synchronize()
    kept = (want4[1].float() > -float("inf")).sum(-1)
    assert int(kept.min()) >= 1 and int(kept.max()) < 32768      # top-p did filter (the failure mode was: silently not applied)
