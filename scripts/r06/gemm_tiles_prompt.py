"""Tile / split-K sweep of the GEMMs of a 2B prompt (M = argv[1], default 386 rows): vlm_gemm_set_staging(100 + 10 * splits + cfg)."""
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

M = int(sys.argv[1]) if len(sys.argv) > 1 else 386
shapes = [("qkv", 2048, 1536, ops.EPI_BIAS), ("o", 1536, 1536, ops.EPI_RESIDUAL), ("gate_up", 17920, 1536, ops.EPI_SWIGLU),
          ("down", 1536, 8960, ops.EPI_RESIDUAL)]
modes = [0, 101, 102, 103, 121, 141, 181]
for name, N, K, epi in shapes:
    ws = [(torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(28)]
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    bias = torch.randn(N, device="cuda").to(torch.bfloat16) if epi & ops.EPI_BIAS else None
    n_out = N // 2 if epi & ops.EPI_SWIGLU else N
    res = torch.randn(M, n_out, device="cuda").to(torch.bfloat16) if epi & ops.EPI_RESIDUAL else None
    out = torch.empty(M, n_out, dtype=torch.bfloat16, device="cuda")
    line = f"{name:8s} M={M} N={N} K={K}:"
    for mode in modes:
        ops.gemm_set_staging(mode)
        it = [0]
        def f():
            it[0] += 1
            ops.gemm(a, ws[it[0] % 28], bias=bias, res=res, out=out, epilogue=epi)
        try:
            line += f"  [{mode}] {ev(f)*1e6:6.1f}"
        except Exception as e:
            line += f"  [{mode}] err"
    print(line)
ops.gemm_set_staging(0)
