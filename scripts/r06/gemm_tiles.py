"""Tile / split-K sweep of the plain GEMM kernels on the single-image ViT shapes (M = 1024 patches of one 448^2 image) and the
prompt shapes; vlm_gemm_set_staging(100 + 10 * splits + cfg), cfg 1 = 64x64, 2 = 64x128, 3 = 128x128; 0 = the automatic policy."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mlx_vlm_amd import ops

def ev(fn, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn(); torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [("qkv", 3840, 1280, ops.EPI_BIAS), ("proj", 1280, 1280, ops.EPI_BIAS | ops.EPI_RESIDUAL), ("fc1", 5120, 1280, ops.EPI_BIAS | ops.EPI_GELU_FAST),
          ("fc2", 1280, 5120, ops.EPI_BIAS | ops.EPI_RESIDUAL)]
modes = [0, 101, 102, 103, 121, 131, 141, 161, 181]
for name, N, K, epi in shapes:
    # rotate over 8 weight copies so the weights come from HBM / MALL as in the tower (32 blocks)
    ws = [(torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(8)]
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16) if epi & ops.EPI_RESIDUAL else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    line = f"{name:5s} M={M} N={N} K={K}:"
    for mode in modes:
        ops.gemm_set_staging(mode)
        it = [0]
        def f():
            it[0] += 1
            ops.gemm(a, ws[it[0] % 8], bias=bias, res=res, out=out, epilogue=epi)
        try:
            dt = ev(f)
            line += f"  [{mode}] {dt*1e6:6.1f}"
        except Exception as e:
            line += f"  [{mode}] err"
    print(line)
ops.gemm_set_staging(0)
