"""The sampler filters beyond top-k / top-p / min-p (top-n-sigma, p-less, locally typical, XTC, min_tokens_to_keep): the
oracle's restatements (oracle/ops.py) against the reference's own functions executed over the shim
(tests/golden/make_golden_samplers.py -> samplers_ref.npz), bit for bit - which tokens survive and their values."""
import os

import numpy as np
import pytest
import torch

from oracle import ops as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "samplers_ref.npz"))


def _same(got, ref):
    got = got.to(torch.float32).numpy()
    assert np.array_equal(np.isfinite(got), np.isfinite(ref)), (np.isfinite(got).sum(-1), np.isfinite(ref).sum(-1))
    assert np.array_equal(got[np.isfinite(ref)], ref[np.isfinite(ref)])


@pytest.mark.parametrize("tag", ["bf16", "f32"])
def test_exotic_sampler_filters_match_the_reference(tag):
    dt = torch.bfloat16 if tag == "bf16" else torch.float32
    x = torch.from_numpy(G[f"{tag}.logprobs"]).to(dt)
    for ns in (0.5, 1.5):
        _same(O.apply_top_n_sigma(x, ns), G[f"{tag}.top_n_sigma_{ns}"])
    for temp in (0.7, 1.3):
        _same(O.apply_p_less(x, temp), G[f"{tag}.p_less_{temp}"])
    for tp in (0.3, 0.9):
        _same(O.apply_typical_p(x, tp), G[f"{tag}.typical_p_{tp}"])
    sp = [int(v) for v in G["xtc_special"]]
    for thr in (0.02, 0.08):
        got = torch.cat([O.apply_xtc(x[r:r + 1], True, thr, sp) for r in range(x.shape[0])])
        _same(got, G[f"{tag}.xtc_{thr}"])
        _same(torch.cat([O.apply_xtc(x[r:r + 1], False, thr, sp) for r in range(x.shape[0])]), G[f"{tag}.xtc_{thr}_never"])
    for mp, keep in ((0.3, 4), (0.05, 1), (0.9, 7)):
        _same(O.apply_min_p(x, mp, keep), G[f"{tag}.min_p_{mp}_keep_{keep}"])
    for tp in (0.5, 0.9, 0.99):
        _same(O.apply_top_p(x, tp), G[f"{tag}.top_p_{tp}"])


@pytest.mark.parametrize("tag", ["bf16", "f32"])
def test_filter_chains_follow_make_samplers_own_order(tag):
    """`chain_<i>`: the row the reference's make_sampler closure hands to its draw (the draw replaced by the identity in
    make_golden_samplers.py) - the oracle's sampler_filters applies the same filters in the same order."""
    import json

    chains = json.loads(str(G["chains_json"]))
    dt = torch.bfloat16 if tag == "bf16" else torch.float32
    x = torch.from_numpy(G[f"{tag}.logprobs"]).to(dt)
    for ci, kw in enumerate(chains):
        if kw.get("xtc_probability"):
            got = torch.cat([O.sampler_filters(x[r:r + 1], 0.8, **kw) for r in range(x.shape[0])])
        else:
            got = O.sampler_filters(x, 0.8, **kw)
        _same(got, G[f"{tag}.chain_{ci}"])
