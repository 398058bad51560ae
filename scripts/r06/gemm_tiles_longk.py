"""Long-K GEMMs with 256..512 tiles of 64 x 64 (not split-K under the t64 < 256 rule): automatic policy vs forced splits."""
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

shapes = [("vit fc2 1 img 448", 1024, 1280, 5120), ("vit fc2 2 img 336", 1152, 1280, 5120), ("7b down T=386", 386, 3584, 18944),
          ("7b down T=640", 640, 3584, 18944), ("2b down T=1024", 1024, 1536, 8960), ("idefics down T=386", 386, 4096, 14336),
          ("7b down T=1024", 1024, 3584, 18944), ("7b down T=1536", 1536, 3584, 18944), ("idefics down T=1024", 1024, 4096, 14336),
          ("7b o_proj T=640", 640, 3584, 3584), ("7b qkv T=640", 640, 4608, 3584), ("7b o_proj T=1024", 1024, 3584, 3584)]
modes = [0, 101, 102, 131]
for name, M, N, K in shapes:
    ws = [(torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(4)]
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    line = f"{name:20s} M={M} N={N} K={K} t64={((M+63)//64)*((N+63)//64)}:"
    for mode in modes:
        ops.gemm_set_staging(mode)
        it = [0]
        def f():
            it[0] += 1
            ops.gemm(a, ws[it[0] % 4], res=res, out=out, epilogue=ops.EPI_RESIDUAL)
        line += f"  [{mode}] {ev(f)*1e6:6.1f}"
    print(line)
ops.gemm_set_staging(0)
