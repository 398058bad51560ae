"""rocprofv3 target: vlm_sample_ex calls at V = 151,936 (one row) per filter configuration, 20 launches each - the kernel
durations of sample_filter_kernel land in the kernel trace in this order (scripts/r04_gpu25.sh prints them per group)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_vlm_amd import ops  # noqa: E402

V = 151936
x = (torch.randn(1, V) * 2).to(torch.bfloat16).cuda()
step = torch.zeros(1, dtype=torch.int32, device="cuda")
ws = ops.sample_workspace(1, x.device)
CASES = [("top_p", dict(top_p=0.9)), ("top_k", dict(top_k=50)), ("min_p", dict(min_p=0.05)),
         ("classic chain", dict(top_p=0.9, min_p=0.02, top_k=50)), ("top_n_sigma", dict(top_n_sigma=1.0)), ("p_less", dict(p_less=True)),
         ("typical_p", dict(typical_p=0.9)), ("xtc", dict(xtc_probability=1.0, xtc_threshold=0.01)),
         ("min_keep", dict(min_p=0.5, min_tokens_to_keep=100)), ("plain", {})]     # (plain: no filter launch - last)
for name, kw in CASES:
    for _ in range(20):
        ops.sample(x, temperature=0.8, step=step, want_logprobs=False, ws=ws, **kw)
    torch.cuda.synchronize()
print("cases:", ",".join(n for n, _ in CASES))
