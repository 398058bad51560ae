#!/usr/bin/env python
"""Fixed cost vs per-K-tile cost of the 256-tile GEMM on the ViT shapes: time(K) = a + b * K/64, fitted per (N, epilogue)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_vlm_amd import ops

def ev(fn, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

M = 9216
for N in (1280, 3840, 5120):
    for epi_name in ("none", "bias+res"):
        if epi_name == "bias+res" and N != 1280:
            continue
        ks, ts = [64, 128, 256, 640, 1280, 2560, 5120], []
        for K in ks:
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
            out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            if epi_name == "none":
                t = ev(lambda: ops.gemm(a, w, out=out))
            else:
                b = torch.randn(N, device="cuda").to(torch.bfloat16)
                r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
                t = ev(lambda: ops.gemm(a, w, bias=b, res=r, out=out, epilogue=ops.EPI_BIAS | ops.EPI_RESIDUAL))
            ts.append(t)
        kt = np.array(ks) / 64
        bfit, afit = np.polyfit(kt[2:], np.array(ts)[2:], 1)
        print(f"N={N:5d} {epi_name:9s} " + " ".join(f"K{k}:{t:6.1f}" for k, t in zip(ks, ts)) + f"   fit: fixed {afit:5.1f} us + {bfit:5.2f} us per K-tile")
