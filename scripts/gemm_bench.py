#!/usr/bin/env python
"""GEMM micro-benchmark (HIP events): TFLOP/s of vlm_gemm_bf16 on the ViT / prefill shapes and 4096^3.
usage: gemm_bench.py [staging modes ...]   (0 auto, 1 register-staged 128 kernel, 2 LDS-DMA 128 kernel, 3 256x256 phased,
4 its 2-phase variant, 6 / 7 256x192 / 256x256 tiles forced; 11-13 = ablation probes, only in a library built with -DVLM_GEMM_ABLATION)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_vlm_amd import ops

def ev(fn, reps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps

shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (9216, 3840, 1280), (9216, 1280, 1280), (9216, 5120, 1280), (9216, 1280, 5120),
          (1024, 3840, 1280), (1024, 5120, 1280), (1024, 1280, 5120), (386, 17920, 1536), (386, 1536, 8960)]
if os.environ.get("GEMM_SHAPES") == "vit":      # the ViT block's GEMMs at 16 and 64 images of 336 x 336 per call
    shapes = [(9216, 3840, 1280), (9216, 5120, 1280), (36864, 3840, 1280), (36864, 5120, 1280), (36864, 1280, 1280), (36864, 1280, 5120)]
EPI = {"none": 0, "bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS | ops.EPI_GELU_FAST}[os.environ.get("GEMM_EPI", "none")]
modes = [int(x) for x in sys.argv[1:]] or [0]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    line = f"{M:6d} {N:6d} {K:6d}"
    for mode in modes:
        ops.gemm_set_staging(mode)
        bias = torch.randn(N, device="cuda").to(torch.bfloat16) if EPI else None
        dt = ev(lambda: ops.gemm(a, w, out=out, bias=bias, epilogue=EPI) if EPI else ops.gemm(a, w, out=out))
        line += f"  mode{mode}: {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:7.1f} TF"
    print(line)
ops.gemm_set_staging(0)
