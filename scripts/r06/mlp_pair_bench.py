#!/usr/bin/env python
"""the MLP of a batched decode step at 2B dims: plain pair vs the tiled hand-over (row-slice down projection), rotating weight copies
larger than the Infinity Cache, launches captured in one graph (the method of scripts/mfma_shapes.py)
usage: mlp_pair_bench.py [rows ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mlx_vlm_amd import ops
BF = torch.bfloat16
def timed(fn, n_copies, reps=40):
    for i in range(n_copies): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for r in range(reps): fn(r % n_copies)
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best
for name, H, I in (("2b", 1536, 8960), ("0.5b", 1024, 2816 * 2), ("3b", 2048, 11008)):
    if I % 128: continue
    n = 12
    wgu = [torch.empty(2 * I, H, dtype=BF, device="cuda").normal_(0, 0.05) for _ in range(n)]
    wd = [torch.empty(H, I, dtype=BF, device="cuda").normal_(0, 0.03) for _ in range(n)]
    nw = torch.ones(H, dtype=BF, device="cuda")
    for M in [int(a) for a in sys.argv[1:]] or [16, 8]:
        x = torch.randn(M, H, device="cuda").to(BF)
        h = torch.randn(M, H, device="cuda").to(BF)
        act = torch.empty(M, I, dtype=BF, device="cuda")
        act_t = torch.zeros(I // 8, 16, 8, dtype=BF, device="cuda")
        t_gu = timed(lambda i: ops.gemv_ws(x, wgu[i], norm_w=nw, out=act, epilogue=ops.EPI_SWIGLU), n)
        t_gu_t = timed(lambda i: ops.gemv_ws(x, wgu[i], norm_w=nw, out=act_t, epilogue=ops.EPI_SWIGLU | ops.EPI_Y_TILED), n)
        t_d = timed(lambda i: ops.gemv_ws(act, wd[i], res=h, out=h, epilogue=ops.EPI_RESIDUAL), n)
        t_d_t = timed(lambda i: ops.gemv_ws(act_t, wd[i], res=h, out=h, epilogue=ops.EPI_RESIDUAL | ops.EPI_X_TILED, M=M), n)
        gb = lambda us, b: b / us / 1e3
        print(f"{name} rows {M:2d}: gate/up {t_gu:6.2f} us (tiled out {t_gu_t:6.2f})   down {t_d:6.2f} us = {gb(t_d, H*I*2):.0f} GB/s -> "
              f"row-slice {t_d_t:6.2f} us = {gb(t_d_t, H*I*2):.0f} GB/s", flush=True)
    del wgu, wd
    torch.cuda.empty_cache()
