"""Timeline + timing of the fused decode block (csrc/decode_block.hip) at Qwen2-VL-2B dims on MI355X.

28 "layers" of weights (60 MB each: HBM-cold in rotation), one captured graph of 28 launches per arm:
  A  three launches per layer: page-split attention (partials) -> merge + o_proj + residual -> RMSNorm + gate/up + SwiGLU
  B  ONE fused launch per layer (+ the epoch bump kernel this standalone call needs; the engine bumps inside the qkv launch)
and the wall-clock stamps (10 ns ticks) of the last fused launch: workgroup 0 (an attention unit), workgroup 255 (o_proj rows
+ gate/up rows).

    python scripts/block_stamps.py [ctx=450] [reps=20]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mlx_vlm_amd import ops

BF = torch.bfloat16
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 450
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
Hq, Hkv, D, inter, S, NL = 12, 2, 128, 8960, 16, 28
K = Hq * D
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(BF)   # noqa: E731
max_pages = 64
kpool = rn(NL, max_pages, Hkv, D // 8, 64, 8)
vpool = rn(NL, max_pages, Hkv, D, 64)
wo = [rn(K, K, sc=0.03) for _ in range(NL)]
wgu = [rn(2 * inter, K, sc=0.03) for _ in range(NL)]
ln2 = torch.ones(K, dtype=BF, device=dev)
q = rn(1, K)
h0 = rn(1, K)
kv_len = torch.tensor([ctx], dtype=torch.int32, device=dev)
scale, eps = D ** -0.5, 1e-6
assert ops.decode_block_supported(Hq, Hkv, D, inter, S), "fused block not supported here"

h = h0.clone()
act = torch.empty(1, inter, dtype=BF, device=dev)
part = (torch.zeros(1, Hq, S, D, dtype=BF, device=dev), torch.zeros(1, Hq, S, 2, dtype=torch.float32, device=dev))
ws = None


def arm_a():
    for l in range(NL):
        po, pml = ops.attn_decode_paged_split(q, kpool[l], vpool[l], None, kv_len, 0, Hq, Hkv, D, scale, S, max_pages=max_pages, merge=False, part=part)
        ops.gemv_attn_out_bf16_(po, pml, wo[l], h, Hq, D)
        ops.gemv(h, wgu[l], norm_w=ln2, eps=eps, epilogue=ops.EPI_SWIGLU, out=act)


def arm_b(stamps=False):
    global ws
    for l in range(NL):
        _, _, ws = ops.decode_block_(q, kpool[l], vpool[l], None, kv_len, 0, Hq, Hkv, D, scale, S, wo[l], h, ln2, eps, wgu[l], act, ws=ws,
                                     max_pages=max_pages, stamps=stamps, part=part)


def timed(fn, name):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()                                   # warm (allocations of the partial buffers happen here, outside the capture)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps / NL
    print(f"{name:58s} {us:7.2f} us per layer")
    return us


a = timed(arm_a, "A  attention | merge + o_proj | norm + gate/up (3 launches)")
b = timed(arm_b, "B  fused decode block (1 launch + epoch bump)")
a2 = timed(arm_a, "A  again")
b2 = timed(arm_b, "B  again")
arm_b(stamps=True)
torch.cuda.synchronize()
err, st = ops.decode_block_debug(ws)
print("hand-offs that gave up:", err)
t0 = min(x for x in st[:12] if x)
us = [(x - t0) / 100.0 if x else float("nan") for x in st]
print(f"workgroup 0 (attention unit g0 s0): start {us[0]:.2f} | partial published {us[1]:.2f} | kv head's partials seen {us[2]:.2f} | "
      f"slice published {us[3]:.2f} | x in LDS {us[11]:.2f}")
print(f"an o_proj wave (layout 1: workgroup NU-1 wave 0; layout 0: workgroup 255 wave 7): start {us[4]:.2f} | attention vector gathered {us[5]:.2f} | rows published {us[6]:.2f} | "
      f"x in LDS {us[7]:.2f}")
print(f"workgroup 255, wave 0 (gate/up rows; its wave 7's 'x in LDS' is the line above): weights requested {us[8]:.2f} | past the barrier {us[9]:.2f} | end {us[10]:.2f}")
