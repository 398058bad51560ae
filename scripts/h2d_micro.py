"""Where does the host time of one pixel upload go?  2.7 MB fp32 (one 336 x 336 image) pageable -> pinned ring -> device."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.init()
src = [torch.randn(576, 1176) for _ in range(8)]
pin = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
plain = torch.empty(64 << 20, dtype=torch.uint8)
dev = torch.empty(8, 576, 1176, device="cuda")
n = src[0].numel() * 4


def T(fn, reps=5):
    ts = []
    for r in range(reps):
        t0 = time.perf_counter()
        fn(r)
        ts.append(1e3 * (time.perf_counter() - t0))
    return " ".join(f"{t:.2f}" for t in ts)


print("threads", torch.get_num_threads())
print("pageable -> pageable   copy_:", T(lambda r: plain[r * n:(r + 1) * n].view(torch.float32).view(576, 1176).copy_(src[r])))
print("pageable -> pinned     copy_:", T(lambda r: pin[r * n:(r + 1) * n].view(torch.float32).view(576, 1176).copy_(src[r])))
print("pinned   -> device  enqueue :", T(lambda r: dev[r].copy_(pin[r * n:(r + 1) * n].view(torch.float32).view(576, 1176), non_blocking=True)))
torch.cuda.synchronize()
print("pageable -> device (sync)   :", T(lambda r: dev[r].copy_(src[r])))
import numpy as np
a = src[0].numpy()
b = np.frombuffer(pin.numpy(), dtype=np.float32, count=576 * 1176).reshape(576, 1176)
print("numpy copyto into pinned    :", T(lambda r: np.copyto(b, a)))
torch.set_num_threads(1)
print("1 thread pageable -> pinned :", T(lambda r: pin[r * n:(r + 1) * n].view(torch.float32).view(576, 1176).copy_(src[r])))
