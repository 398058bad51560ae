"""Phase timeline of sample_filter_kernel (csrc/sample.hip built with -DVLM_SAMPLE_STAMPS into scripts/bin/libsample_stamps.so:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVLM_SAMPLE_STAMPS -mllvm -amdgpu-mfma-vgpr-form \
          mlx-vlm_amd/csrc/sample.hip -o scripts/bin/libsample_stamps.so
): one row of V = 151,936, the 100 MHz wall clock at the phase boundaries of the filter kernel (the draw is its own launch)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mlx_vlm_amd import _lib  # noqa: E402

L = C.CDLL(os.path.join(ROOT, "scripts", "bin", "libsample_stamps.so"))
L.vlm_sample_workspace_bytes.restype = C.c_size_t
L.vlm_sample_ex.argtypes = _lib.SIGNATURES["vlm_sample_ex"][1]
V = 151936
x = (torch.randn(1, V) * 2).to(torch.bfloat16).cuda()
lp = torch.empty_like(x)
scratch = torch.empty_like(x)
tok = torch.zeros(1, dtype=torch.int32, device="cuda")
ws = torch.zeros(L.vlm_sample_workspace_bytes(1), dtype=torch.uint8, device="cuda")
step = torch.zeros(1, dtype=torch.int32, device="cuda")
NAMES = {True: ["copy + earlier filters", "histogram", "mass + scan", "crossing walk", "rank in bin", "mask", "later filters"],
         False: ["copy + earlier filters", "later filters (top-k: histogram, walk, rank, mask)"]}
for name, kw in (("top_p 0.9", dict(top_p=0.9)), ("top_p 0.5", dict(top_p=0.5)), ("plain", dict()), ("top_k 50", dict(top_k=50))):
    sp = _lib.SamplerParams(temperature=0.8, top_p=kw.get("top_p", 0.0), min_p=0.0, min_tokens_to_keep=1, top_k=kw.get("top_k", 0),
                            typical_p=1.0, seed=3)
    for _ in range(3):
        rc = L.vlm_sample_ex(x.data_ptr(), V, 1, V, lp.data_ptr(), scratch.data_ptr(), V, tok.data_ptr(), ws.data_ptr(), C.byref(sp),
                             step.data_ptr(), None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    h = ws[256 + 64 * 4 * 4:256 + 64 * 4 * 4 + 128].cpu().numpy().view(np.uint32)
    n = int(h[31])
    t = h[:n].astype(np.int64)
    d = (np.diff(t) & 0xFFFFFFFF) / 100.0
    names = NAMES["top_p" in kw]
    print(f"{name}: total {d.sum():.1f} us  " + "  ".join(f"{names[i] if i < len(names) else i}: {v:.1f}" for i, v in enumerate(d)))
