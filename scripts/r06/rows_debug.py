import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mlx_vlm_amd import ops
BF = torch.bfloat16
torch.manual_seed(0)
H, I = 1536, 8960
for M, fill in ((16, 0.0), (5, 0.0), (5, float("nan"))):
    a = (torch.randn(M, I) * 0.3).to(BF).cuda()
    wd = (torch.randn(H, I) * 0.03).to(BF).cuda()
    at = ops.tile_rows(a)
    if M < 16: at[:, M:] = fill
    ref = (a.double() @ wd.double().T)
    got = ops.gemv_ws(at, wd, epilogue=ops.EPI_X_TILED, M=M).double()
    plain = ops.gemv_ws(a, wd).double()
    err = (got - ref).abs()
    print(f"M={M} fill={fill}: rows-form max err {float(err.nan_to_num(1e9).max()):.4f} (plain {float((plain-ref).abs().max()):.4f}); nan count {int(torch.isnan(got).sum())}")
    bad = (err.nan_to_num(1e9) > 0.1)
    print("  bad per batch row:", bad.sum(1).tolist())
    print("  bad per (n % 6):", [int(bad[:, r::6].sum()) for r in range(6)])
    print("  sample got/ref:", got[0, :8].tolist(), ref[0, :8].tolist())
