"""ViT attention launch time (16 / 64 images x 576 patches x 16 heads x 80) and, with CAUSAL=1, the LLM prompt shape; VLM_ATTN_PIPE picks the kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlx_vlm_amd import ops

def ev(fn, reps=50):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

for nimg, L, H, D, causal in ((16, 576, 16, 80, False), (64, 576, 16, 80, False), (8, 729, 16, 80, False), (16, 577, 16, 64, False), (1, 4096, 12, 128, True)):
    T = nimg * L
    qkv = (torch.randn(T, 3 * H * D, device="cuda") * 0.5).to(torch.bfloat16)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    nqb = nimg * ((L + 127) // 128)
    fn = lambda: ops.attn_prefill(qkv, qkv[:, H * D:], qkv[:, 2 * H * D:], cu, nqb, H, H, D, D ** -0.5, causal, uniform_segments=True)
    out = fn()
    us = ev(fn)
    fl = 4.0 * nimg * L * L * H * D * (0.5 if causal else 1.0)
    print(f"pipe={os.environ.get('VLM_ATTN_PIPE','default')} imgs={nimg} L={L} H={H} D={D} causal={causal}: {us:8.2f} us  {fl / us / 1e6:7.1f} TF  checksum {float(out.float().abs().sum()):.6e}")
