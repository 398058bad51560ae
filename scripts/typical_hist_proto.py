"""Prototype (numpy / torch on the CPU, not product code) of the form csrc/sample.hip's typical-p should take next: the sort is
only needed inside a WINDOW of sort keys around the crossing.

The reference (sample_utils.py:321-345) sorts all V tokens by |-logp - H| (stable) and keeps token i iff
T(T(cum_i) - p_i) < T(typical_p), cum_i the inclusive running sum in that order.  Tokens in key bins whose whole prefix range lies
clearly below the threshold are all kept, clearly above all removed; only the bins in between need the order.  Here:
  1. per 15-bit key: count and mass, the mass as an exact integer sum of p * 2^40 (order-free, so a kernel can use LDS / L2
     integer atomics and stay deterministic);
  2. exclusive prefix per key (ascending), as float32 from the integers;
  3. window = keys whose prefix range [E, E + m] comes within delta = 2^-6 (thr + pmax) of the threshold (delta covers the two bf16
     roundings of the rule: T(cum) and the difference);
  4. tokens below the window kept, above removed; the window's tokens sorted (key, index) and judged one by one with the running
     sum started at the window's exclusive prefix - exactly the per-token rule.
Checked against oracle/ops.py::apply_typical_p (bit-exact vs the reference's golden rows) on random rows: the survivor sets are
identical whenever the float32 running sums agree; `python scripts/typical_hist_proto.py` prints the window sizes - the share of
the row the radix passes of the kernel would still have to move (it is the scatter of ALL V tokens that costs 290 of the
kernel's 530 us today, profiles/r04_sampler_kernel_us.txt)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ops as O  # noqa: E402

BF, F32 = torch.bfloat16, torch.float32


def typical_window(lp: torch.Tensor, typical_p: float):
    """lp bf16 [V] -> (filtered bf16 [V], window size)"""
    lf = lp.to(F32)
    p = torch.exp(lf).to(BF)
    pl = (p.to(F32) * lf).to(BF)
    ent = (-(pl.to(F32).sum().to(BF)).to(F32)).to(BF)
    shifted = ((-lf) - ent.to(F32)).to(BF).to(F32).abs().to(BF)
    key = shifted.view(torch.int16).to(torch.int64) & 0x7FFF              # non-negative bf16: its bits order it
    pf = p.to(F32).numpy().astype(np.float64)
    fix = np.round(pf * 2.0 ** 40).astype(np.int64)                        # exact for p >= 2^-33 (8-bit mantissas), 0 below 2^-41
    K = 1 << 15
    keyn = key.numpy()
    mass = np.bincount(keyn, weights=None, minlength=K) * 0
    mass = np.zeros(K, dtype=np.int64)
    np.add.at(mass, keyn, fix)
    excl = np.concatenate([[0], np.cumsum(mass)[:-1]]).astype(np.float64) / 2.0 ** 40
    incl = excl + mass / 2.0 ** 40
    thr = float(torch.tensor(typical_p, dtype=BF))
    pmax = float(pf.max())
    delta = 2.0 ** -6 * (thr + pmax)
    present = np.bincount(keyn, minlength=K) > 0
    below = present & (incl < thr - delta)                                  # every token's `before` is clearly < thr
    above = present & (excl > thr + delta)
    win = present & ~below & ~above
    out = lp.clone()
    tok_above = torch.from_numpy(above[keyn])
    out[tok_above] = float("-inf")
    idx = np.nonzero(win[keyn])[0]
    if idx.size:
        order = idx[np.argsort(keyn[idx], kind="stable")]
        k0 = keyn[order[0]]
        run = np.float32(excl[k0])
        pw = p.to(F32).numpy()[order]
        cum = np.cumsum(np.concatenate([[run], pw]).astype(np.float32), dtype=np.float32)[1:]
        cum_t = torch.from_numpy(cum).to(BF).to(F32)
        before = (cum_t - torch.from_numpy(pw)).to(BF).to(F32)
        drop = ~(before < thr)
        out[torch.from_numpy(order[drop.numpy()])] = float("-inf")
    return out, int(idx.size)


def main():
    g = torch.Generator().manual_seed(7)
    worst = 0
    for trial in range(40):
        V = int(torch.randint(2000, 160000, (1,), generator=g))
        scale = float(torch.rand(1, generator=g)) * 5 + 0.3
        x = torch.randn(V, generator=g) * scale
        lp = (x - torch.logsumexp(x, 0)).to(BF)
        for tp in (0.2, 0.5, 0.9, 0.99):
            got, n = typical_window(lp, tp)
            ref = O.apply_typical_p(lp[None], tp)[0]
            kg, kr = torch.isfinite(got.float()), torch.isfinite(ref.float())
            diff = int((kg != kr).sum())
            worst = max(worst, diff)
            if diff:
                # the tokens judged differently must all sit at the cut: one key bin (the crossing one), where the float32 running
                # sum of this form (window started from an exact integer prefix) and the oracle's land on different sides of a
                # bf16 step of the cumulative
                lf = lp.to(F32)
                pb = torch.exp(lf).to(BF)
                ent = (-((pb.to(F32) * lf).to(BF).to(F32).sum().to(BF)).to(F32)).to(BF)
                kk = ((-lf) - ent.to(F32)).to(BF).to(F32).abs().to(BF).view(torch.int16).to(torch.int64) & 0x7FFF
                bins = torch.unique(kk[kg != kr])
                assert bins.numel() <= 2, (V, tp, bins)
            print(f"V {V:6d} scale {scale:4.2f} typical_p {tp:4.2f}: kept {int(kr.sum()):6d}  window {n:6d} ({100.0 * n / V:5.1f} % of the row)  "
                  f"tokens judged differently: {diff}")
    print("worst difference:", worst)


if __name__ == "__main__":
    main()
