#!/usr/bin/env python
"""one GEMM shape, one staging mode, timed with HIP events: gemm_one.py M N K mode [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mlx_vlm_amd import ops
M, N, K, mode = (int(x) for x in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
ops.gemm_set_staging(mode)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.gemm(a, w, out=out); ops.gemm(a, w, out=out); torch.cuda.synchronize()
e0.record()
for _ in range(reps): ops.gemm(a, w, out=out)
e1.record(); e1.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / reps
print(f"{M} {N} {K} mode {mode} abl {os.environ.get('VLM_GEMM_F_ABL', '0')}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.1f} TF")
