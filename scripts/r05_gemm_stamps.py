"""Timeline of the 256-wide GEMM's workgroups (library built with -DGEMM_STAMPS, see csrc/gemm256_bf16.hip): where does the fixed cost
per round of tiles go - workgroup turnover on a CU, the first LDS-DMA round trip, the K loop, the epilogue's conversion, its stores?
usage: VLM_HIP_LIB=.../libvlm_hip_stamps.so python scripts/r05_gemm_stamps.py [M N K [epi]]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mlx_vlm_amd import ops, _lib

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (36864, 5120, 1280)
epi = sys.argv[4] if len(sys.argv) > 4 else "gelu"
EPI = {"none": 0, "bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS | ops.EPI_GELU_FAST}[epi]
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda").to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
ops.gemm_set_staging(3)
run = lambda: ops.gemm(a, w, out=out, bias=bias if EPI else None, epilogue=EPI)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
L = _lib.lib()
tn = 256
ntiles = ((M + 255) // 256) * ((N + tn - 1) // tn)
ntiles192 = ((M + 255) // 256) * ((N + 191) // 192)
n = min(16384, max(ntiles, ntiles192))
buf = np.zeros((n, 8), dtype=np.uint64)
L.vlm_debug_gemm_stamps.restype = C.c_int
rc = L.vlm_debug_gemm_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(n))
assert rc == 0, rc
st = buf[buf[:, 0] > 0]
t = st[:, [0, 1, 2, 3, 6, 4]].astype(np.float64) * 0.01        # 100 MHz ticks -> us
t0 = t[:, 0].min()
t -= t0
hw = st[:, 5]
xcc = (hw >> np.uint64(32)).astype(np.int64)
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7       # HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
key = xcc * 1000 + se * 100 + sh * 20 + cu
print(f"{M} x {N} x {K} epi={epi}: {len(st)} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (events), last end {t[:, 5].max():.1f} us, distinct CU keys {len(set(key.tolist()))}")
names = ["entry->first K tile landed", "K loop", "epilogue: convert half A + sync", "convert half B + stores A + sync", "stores B"]
seg = np.diff(t, axis=1)
for i, nm in enumerate(names):
    print(f"  {nm:38s} avg {seg[:, i].mean():6.2f}  p10 {np.percentile(seg[:, i], 10):6.2f}  median {np.median(seg[:, i]):6.2f}  p90 {np.percentile(seg[:, i], 90):6.2f} us")
print(f"  {'workgroup total':38s} avg {(t[:, 5] - t[:, 0]).mean():6.2f} us")
# turnover: per CU, the gap between one workgroup's end and the next one's entry
gaps, per_cu = [], []
for k in set(key.tolist()):
    idx = np.where(key == k)[0]
    o = idx[np.argsort(t[idx, 0])]
    per_cu.append(len(o))
    for a_, b_ in zip(o[:-1], o[1:]):
        gaps.append(t[b_, 0] - t[a_, 5])
gaps = np.array(gaps)
print(f"  workgroups per CU: min {min(per_cu)} max {max(per_cu)}; turnover (end of one workgroup -> entry of the next on the same CU): "
      f"avg {gaps.mean():.2f}  p10 {np.percentile(gaps, 10):.2f}  median {np.median(gaps):.2f}  p90 {np.percentile(gaps, 90):.2f} us  (n = {len(gaps)})")
first = np.sort(t[:, 0])
print(f"  entries: first 256 workgroups enter within {first[min(255, len(first) - 1)]:.2f} us of the first; per-round entry medians:",
      " ".join(f"{np.median(first[i:i + 256]):.1f}" for i in range(0, len(first), 256)))
