#!/usr/bin/env python
"""How far apart are two CORRECT evaluations of the reference's bf16 typed graph that differ only in the ORDER / precision in which
the fp32 sums inside a matmul are accumulated - as a function of depth?  (round-4 review, item 5: the full-depth bars of
tests/test_full_depth_gpu.py are 4e-2 rel-rms on a logit row for configs[1]; is that a loose bar or the floor of the comparison?)

Both runs below are the oracle (oracle/qwen2_vl.py, bit-exact against the reference's own files on the tiny model): same weights,
same inputs, same rounding points (every materialised tensor rounds to bf16).  Run A accumulates every `nn.Linear` in float32 (the
oracle's statement of MLX: torch's blocked sgemm order), run B in float64 (the exactly-rounded sum) - a third order is what the HIP
kernels have (MFMA 16x16x32 fragments, K tiles of 64, split-K partials).  Qwen2-VL-2B text dims, N(0, 0.02^2) weights (BASELINE's
synthetic checkpoint), a 64-token text prompt, the model truncated to L layers; the distance of the last row's logits.
CPU only; usage: python scripts/r05_depth_noise.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ops as O
from oracle import qwen2_vl as oq

torch.set_num_threads(min(16, os.cpu_count() or 8))
cfg = oq.Cfg()
t = cfg.text
W = oq.random_weights(cfg, seed=0, dtype=torch.bfloat16, fast=True)
ids = np.random.default_rng(1000).integers(0, 151643, (1, 64))
emb = oq.embed_tokens(W, ids)
pos = torch.arange(64)[None, None].expand(3, 1, 64)
lin32 = O.linear


def lin64(x, w, b=None):
    if hasattr(w, "wq"):
        return w.linear(x, b)
    y = x.to(torch.float64) @ w.to(torch.float64).T
    if b is not None:
        y = y + b.to(torch.float64)
    return y.to(O._result_type(x, w))


def run(depth, lin):
    O.linear = lin
    try:
        c = oq.Cfg()
        c.text.num_hidden_layers = depth
        h = oq.qwen2_model(W, c, emb, None, pos)
        return oq.lm_head(W, c, h[:, -1:, :])[0, 0].float()
    finally:
        O.linear = lin32


lines = ["depth   rel-rms(A, B) of the last row's logits   argmax equal   top-2 margin of A / rms"]
for depth in (1, 2, 4, 8, 16, 28):
    t0 = time.time()
    a, b = run(depth, lin32), run(depth, lin64)
    rr = float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    top = a.topk(2).values
    lines.append(f"{depth:5d}   {rr:10.4e}                               {int(a.argmax()) == int(b.argmax())!s:5s}          "
                 f"{float((top[0] - top[1]) / a.pow(2).mean().sqrt()):.3f}      ({time.time() - t0:.0f} s)")
    print(lines[-1], flush=True)
txt = "\n".join(lines)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(__doc__ + "\n" + txt + "\n")
