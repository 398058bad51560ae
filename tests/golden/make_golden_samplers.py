"""Golden vectors for the sampler filters beyond top-k / top-p / min-p ("exotic samplers: next"):
top-n-sigma, p-less, locally typical, XTC and min-p's min_tokens_to_keep - produced by the REFERENCE'S OWN
`mlx_vlm/sample_utils.py` (lines 181-376) imported unmodified from /root/reference and executed over oracle/mlx_shim (see
make_golden_ref.py for the import machinery).  Run once in the build container:

    python tests/golden/make_golden_samplers.py        -> tests/golden/samplers_ref.npz

Inputs are normalised log-probabilities in bf16 (what generate_step hands a sampler, ar.py:368) and in fp32.  XTC draws
`mx.random.uniform(0, 1) > xtc_probability`: recorded at probability 1.0 (always applied) and 0.0 (never).  What is recorded
per case: the filter's output as float32 (-inf where a token was removed).  `chain_<i>`: the row that make_sampler's OWN closure
hands to its draw (CHAINS below; the draw replaced by the identity): the order of the filters.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden_ref as G  # noqa: E402


# filter chains through make_sampler itself (tests read this list back)
CHAINS = [dict(top_p=0.9, min_p=0.05, top_k=20),
          dict(top_n_sigma=2.0, top_p=0.95, top_k=50),
          dict(typical_p=0.9, top_p=0.9, min_p=0.02, min_tokens_to_keep=3, top_k=40),
          dict(p_less=True, top_k=3),
          dict(top_p=0.99, xtc_probability=1.0, xtc_threshold=0.02, xtc_special_tokens=[3, 17], top_k=100),
          dict(min_p=0.5, min_tokens_to_keep=9, top_k=6)]


def main():
    mx, q, cfgm, cache, su = G.import_reference()
    assert su.__file__.startswith(G.REF)
    f32 = lambda a: np.asarray(a.astype(mx.float32)._t.numpy())      # noqa: E731
    g = torch.Generator().manual_seed(123)
    out = {}
    V = 1031
    # three rows: a broad distribution, a peaked one, one with exact ties (quantised logits)
    logits = torch.randn(3, V, generator=g) * torch.tensor([[1.5], [5.0], [2.0]])
    logits[2] = torch.round(logits[2] * 2) / 2
    lp32 = logits - torch.logsumexp(logits, -1, keepdim=True)
    for tag, dt in (("bf16", mx.bfloat16), ("f32", mx.float32)):
        x = mx.array(lp32.numpy()).astype(dt)
        out[f"{tag}.logprobs"] = f32(x)
        for ns in (0.5, 1.5):
            out[f"{tag}.top_n_sigma_{ns}"] = f32(su.apply_top_n_sigma(x, ns))
        for temp in (0.7, 1.3):
            out[f"{tag}.p_less_{temp}"] = f32(su.apply_p_less(x, temp))
        for tp in (0.3, 0.9):
            out[f"{tag}.typical_p_{tp}"] = f32(su.apply_typical_p(x, tp))
        # (apply_xtc takes its threshold token with a min over the WHOLE array: one row per call, as generate_step calls it)
        for thr in (0.02, 0.08):
            out[f"{tag}.xtc_{thr}"] = np.concatenate([f32(su.apply_xtc(x[r:r + 1], 1.0, thr, [3, 17])) for r in range(3)])
            out[f"{tag}.xtc_{thr}_never"] = np.concatenate([f32(su.apply_xtc(x[r:r + 1], 0.0, thr, [3, 17])) for r in range(3)])
        for mp, keep in ((0.3, 4), (0.05, 1), (0.9, 7)):
            out[f"{tag}.min_p_{mp}_keep_{keep}"] = f32(su.apply_min_p(x, mp, keep))
        # the round-1 filters once more on bf16 inputs (qwen2_vl_tiny_ref.npz holds them for fp32 inputs only)
        for tp in (0.5, 0.9, 0.99):
            out[f"{tag}.top_p_{tp}"] = f32(su.apply_top_p(x, tp))
        out[f"{tag}.top_k_5"] = f32(su.apply_top_k(x, 5))
        # the CHAIN: make_sampler's own closure (sample_utils.py:66-89) with its last step - the random draw - replaced by the
        # identity, so that what comes back is the filtered row the draw would see: pins the ORDER of the filters
        keep_draw = su.categorical_sampling
        su.categorical_sampling = lambda lp, temp: lp
        try:
            for ci, kw in enumerate(CHAINS):
                smp = su.make_sampler(temp=0.8, **kw)
                rows = [f32(smp(x[r:r + 1])) for r in range(3)] if kw.get("xtc_probability") else [f32(smp(x))]
                out[f"{tag}.chain_{ci}"] = np.concatenate(rows)
        finally:
            su.categorical_sampling = keep_draw
    out["chains_json"] = np.array(json.dumps(CHAINS))
    out["xtc_special"] = np.array([3, 17], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "samplers_ref.npz"), **out)
    print("wrote samplers_ref.npz:", len(out), "arrays")
    for k in sorted(out):
        if k.endswith("logprobs") or k in ("xtc_special", "chains_json"):
            continue
        a = out[k]
        print(f"  {k:34s} kept per row: {[int(np.isfinite(r).sum()) for r in a]}")


if __name__ == "__main__":
    main()
