#!/usr/bin/env python
"""effective shader clock during the phased 256-wide GEMM (mode 3 / 6 / 7; library built with -DGEMM_STAMPS)
usage: gemm_clock3.py M N K mode [zero]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mlx_vlm_amd import ops, _lib
M, N, K, mode = (int(x) for x in sys.argv[1:5])
zero = len(sys.argv) > 5
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
if zero: a.zero_(); w.zero_()
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
ops.gemm_set_staging(mode)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): ops.gemm(a, w, out=out)
torch.cuda.synchronize(); e0.record()
for _ in range(10): ops.gemm(a, w, out=out)
e1.record(); e1.synchronize()
dt = e0.elapsed_time(e1) * 1e-3 / 10
n = min(16384, ((M + 255) // 256) * ((N + 255) // 256))
buf = np.zeros((n, 8), dtype=np.uint64)
_lib.lib().vlm_debug_gemm_stamps(ctypes.c_void_p(buf.ctypes.data), n)
cyc = buf[:, 7].astype(np.float64); wall = (buf[:, 4] - buf[:, 0]).astype(np.float64) * 10e-9
kl = (buf[:, 2] - buf[:, 1]).astype(np.float64) * 10e-9
ok = wall > 0
print(f"{M} {N} {K} mode {mode} {'zero' if zero else 'randn'}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.1f} TF; workgroup {np.median(wall[ok])*1e6:.1f} us = "
      f"{np.median(cyc[ok])/1e3:.1f} kcycles -> clock {np.median(cyc[ok]/wall[ok])/1e9:.3f} GHz; K loop {np.median(kl[ok])*1e6:.1f} us = "
      f"{np.median(kl[ok])*np.median(cyc[ok]/wall[ok])/(K/64):.0f} cycles per K tile (2048 at the MFMA rate)")
