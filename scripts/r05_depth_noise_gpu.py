#!/usr/bin/env python
"""The HIP engine beside the two oracle evaluations of scripts/r05_depth_noise.py, as a function of depth (round-4 review, item 5:
"measure the per-layer error growth").  Qwen2-VL-2B text dims, N(0, 0.02^2) weights, a 64-token text prompt, the model truncated
to L layers; the last row's logits of the whole-prompt prefill and of ONE decode step after it (a forced token), for

    HIP  the product (flash-attention prefill, MFMA GEMMs; GEMV + page-split attention at the decode step)
    A    the oracle, every nn.Linear accumulated in float32 (its statement of MLX)
    B    the oracle with the sums accumulated in float64 (exactly rounded)

Same weights, same inputs, every materialised tensor rounded to bf16 at the same points in all three.  If the kernels added error
of their own, HIP would sit farther from B than A does.
usage (GPU): python scripts/r05_depth_noise_gpu.py [out.txt]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import ops as O            # (a script under scripts/: the oracle is the checker here, as in tests/)
from oracle import qwen2_vl as oq
from tests.helpers import build_product_model

torch.set_num_threads(min(16, os.cpu_count() or 8))
cfg0 = oq.Cfg()
W = oq.random_weights(cfg0, seed=0, dtype=torch.bfloat16, fast=True)
ids = np.random.default_rng(1000).integers(0, 151643, (1, 64))
FORCED = 4242
lin32 = O.linear


def lin64(x, w, b=None):
    if hasattr(w, "wq"):
        return w.linear(x, b)
    y = x.to(torch.float64) @ w.to(torch.float64).T
    if b is not None:
        y = y + b.to(torch.float64)
    return y.to(O._result_type(x, w))


def oracle_rows(depth, lin):
    O.linear = lin
    try:
        c = oq.Cfg()
        c.text.num_hidden_layers = depth
        return oq.decode_teacher_forced(W, c, ids, None, None, forced_tokens=[FORCED]).float()
    finally:
        O.linear = lin32


def hip_rows(depth):
    c = oq.Cfg()
    c.text.num_hidden_layers = depth
    keep = {k: v for k, v in W.items() if ".layers." not in k or not k.startswith("language_model") or int(k.split(".layers.")[1].split(".")[0]) < depth}
    model = build_product_model(c, keep, kv_pool_tokens=2048, max_seqs=2)
    lm = model.language_model
    f = model.get_input_embeddings(ids, None)
    cache = lm.make_cache()
    out = lm(ids, f.inputs_embeds, cache=cache, position_ids=f.position_ids, rope_deltas=f.rope_deltas, logits_to_keep=1)
    rows = [out.logits[0, -1].float().cpu()]
    rows.append(lm(np.array([[FORCED]]), cache=cache).logits[0, -1].float().cpu())
    cache[0]._seq.release()
    del model
    torch.cuda.empty_cache()
    return torch.stack(rows)


def rr(a, b):
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


lines = ["rel-rms of a logit row (prefill last row | the decode step after it)",
         "depth      HIP vs A            HIP vs B            A vs B            argmax HIP = A = B"]
for depth in (1, 2, 4, 8, 16, 28):
    t0 = time.time()
    h, a, b = hip_rows(depth), oracle_rows(depth, lin32), oracle_rows(depth, lin64)
    same = all(int(h[i].argmax()) == int(a[i].argmax()) == int(b[i].argmax()) for i in range(2))
    lines.append(f"{depth:5d}   {rr(h[0], a[0]):.4e} | {rr(h[1], a[1]):.4e}   {rr(h[0], b[0]):.4e} | {rr(h[1], b[1]):.4e}   "
                 f"{rr(a[0], b[0]):.4e} | {rr(a[1], b[1]):.4e}   {same!s:5s}   ({time.time() - t0:.0f} s)")
    print(lines[-1], flush=True)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(__doc__ + "\n" + "\n".join(lines) + "\n")
