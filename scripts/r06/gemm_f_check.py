#!/usr/bin/env python
"""Round 6: the free-running 256-row GEMM (csrc/gemm256f_bf16.hip, staging modes 14-16) against the 128x128 kernel (mode 2):
agreement to fp32 summation order (<= 1 bf16 ulp, on a small fraction of the elements), ragged edges, every epilogue, repeated
launches as a race screen."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mlx_vlm_amd import ops

BF = torch.bfloat16
torch.manual_seed(0)
bad = 0
def ulps(a, b):
    ia = a.view(torch.int16).to(torch.int32); ib = b.view(torch.int16).to(torch.int32)
    ia = torch.where(ia < 0, -(ia & 0x7fff), ia); ib = torch.where(ib < 0, -(ib & 0x7fff), ib)
    return (ia - ib).abs()
cases = [(256, 256, 128), (512, 768, 256), (1000, 520, 192), (2048, 3840, 1280), (777, 1288, 640), (4096, 4096, 1024), (300, 200, 128)]
epis = {"none": 0, "bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS | ops.EPI_GELU_FAST, "erf": ops.EPI_BIAS | ops.EPI_GELU_ERF,
        "bias_res": ops.EPI_BIAS | ops.EPI_RESIDUAL, "res": ops.EPI_RESIDUAL, "swiglu": ops.EPI_SWIGLU}
for (M, N, K) in cases:
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda").to(BF)
    for name, epi in epis.items():
        if name == "swiglu" and N % 16: continue
        No = N // 2 if name == "swiglu" else N
        res = torch.randn(M, No, device="cuda").to(BF) if epi & ops.EPI_RESIDUAL else None
        kw = dict(bias=bias if epi & ops.EPI_BIAS else None, res=res, epilogue=epi)
        ops.gemm_set_staging(2)
        ref = ops.gemm(a, w, **kw).clone()
        for mode in (15, 16):
            ops.gemm_set_staging(mode)
            for it in range(3):
                out = ops.gemm(a, w, **kw)
                d = ulps(out, ref)
                mx, frac = int(d.max()), float((d > 0).float().mean())
                ok = mx <= 1 and frac < 5e-3 and bool(torch.isfinite(out.float()).all())
                if not ok or it == 0:
                    print(f"{M:5d} {N:5d} {K:5d} {name:8s} mode {mode} it {it}: max ulp {mx} frac {frac:.2e} {'ok' if ok else 'FAIL'}")
                bad += (not ok)
ops.gemm_set_staging(0)
# fp64 truth on one shape: both kernels equally far from it
M, N, K = 1024, 1024, 2048
a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
truth = (a.double() @ w.double().T)
for mode in (2, 3, 16):
    ops.gemm_set_staging(mode)
    o = ops.gemm(a, w).double()
    print(f"mode {mode}: rel-rms vs fp64 {float(((o - truth).pow(2).mean() / truth.pow(2).mean()).sqrt()):.3e}")
ops.gemm_set_staging(0)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
