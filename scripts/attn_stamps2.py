import os, sys
os.environ["VLM_ATTN_STAMPS"] = "1"   # needs a library built with -DVLM_ATTN_TIMELINE (debug timeline stamps)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_vlm_amd import ops
B, Hq, Hkv, D, ctx = 1, 12, 2, 128, 620
npg = (ctx + 63) // 64
kpool = torch.randn(npg + 2, Hkv, D // 8, 64, 8, device="cuda").to(torch.bfloat16)
vpool = torch.randn(npg + 2, Hkv, D, 64, device="cuda").to(torch.bfloat16)
bt = torch.arange(npg + 2, dtype=torch.int32, device="cuda")[None]
kv_len = torch.tensor([ctx], dtype=torch.int32, device="cuda")
q = torch.randn(1, Hq * D, device="cuda").to(torch.bfloat16)
part_o = torch.zeros(B, Hq, 1, D, dtype=torch.float32, device="cuda")
part_ml = torch.zeros(B, Hq, 1, 2, dtype=torch.float32, device="cuda")
out = torch.empty(B, Hq * D, dtype=torch.bfloat16, device="cuda")
import ctypes as C
from mlx_vlm_amd import _lib
L = _lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
def run():
    L.vlm_attn_decode_paged(p(q), q.stride(0), p(kpool), p(vpool), p(bt), bt.shape[1], p(kv_len), 0, B, Hq, Hkv, D, D ** -0.5, 1,
                            p(part_o), p(part_ml), p(out), out.stride(0), None)
names = ["start", "q", "kv issued", "QK", "softmax", "PV", "pre-bar", "post-bar", "end"]
def show(tag):
    torch.cuda.synchronize()
    po = part_o.reshape(-1)[:32].cpu().tolist()
    print(tag, "wave0:", " ".join(f"{names[i]}={po[i]:.2f}" for i in range(9)))
run(); show("cold (first launch)")
for _ in range(20): run()
show("back-to-back x20 (icache warm)")
big.fill_(1); run(); show("after a 256 MB fill (L2 flushed)")
x = torch.randn(4096, 4096, device="cuda"); y = x @ x; run(); show("after an unrelated big kernel")
