"""gate/up of a 2B prompt (SwiGLU, N = 17920, K = 1536) at M around 386: the 256-wide kernel's tile width (mode 6 = 256 x 192, 7 = 256 x 256, 0 = automatic)."""
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

for N, K, epi, name in ((17920, 1536, ops.EPI_SWIGLU, "2b gate/up"), (37888, 3584, ops.EPI_SWIGLU, "7b gate/up")):
    ws = [(torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(6)]
    for M in (256, 384, 386, 448, 512, 576, 640, 770):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        out = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
        line = f"{name} M={M:4d}:"
        for mode in (0, 6, 7):
            ops.gemm_set_staging(mode)
            it = [0]
            def f():
                it[0] += 1
                ops.gemm(a, ws[it[0] % 6], out=out, epilogue=epi)
            line += f"  [{mode}] {ev(f)*1e6:6.1f}"
        print(line)
ops.gemm_set_staging(0)
