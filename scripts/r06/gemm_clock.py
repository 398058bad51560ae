#!/usr/bin/env python
"""effective shader clock during the free-running GEMM (library built with VLM_BUILD_DEFINES including GEMM_STAMPS):
s_memtime (shader cycles) against s_memrealtime (100 MHz) between entry and exit of every workgroup.
usage: gemm_clock.py M N K mode [zero]"""
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
buf = np.zeros((n, 4), dtype=np.uint64)
rc = _lib.lib().vlm_debug_gemmf_stamps(ctypes.c_void_p(buf.ctypes.data), n)
cyc = (buf[:, 2] - buf[:, 0]).astype(np.float64); wall = (buf[:, 3] - buf[:, 1]).astype(np.float64) * 10e-9
print(f"{M} {N} {K} mode {mode} abl {os.environ.get('VLM_GEMM_F_ABL','0')} {'zero' if zero else 'randn'}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.1f} TF; "
      f"workgroup {np.median(wall)*1e6:.1f} us = {np.median(cyc)/1e3:.1f} kcycles -> clock {np.median(cyc/wall)/1e9:.3f} GHz (p10 {np.percentile(cyc/wall,10)/1e9:.3f} p90 {np.percentile(cyc/wall,90)/1e9:.3f})")
