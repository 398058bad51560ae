"""Per-projection time of the batched decode step's GEMVs (csrc/gemv_mfma.hip) at the dims of the benchmark models:
every call site of a decoder layer (qkv + RoPE + KV write, o_proj + residual, RMSNorm + gate/up + SwiGLU, down + residual)
at 8 / 16 rows, each timed over rotating weight copies that together exceed the Infinity Cache (HIP events around N
launches captured in one graph - as in the engine's step, no host cost between the launches).  Policy knobs of the kernel are environment variables read once per process, so an A/B is
two runs of this script.
    python scripts/mfma_shapes.py [2b|7b|mistral|phi-w4 ...] [--rows 16,8] [--reps 40]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mlx_vlm_amd import ops as vops  # noqa: E402

BF = torch.bfloat16
DIMS = {"2b": dict(H=1536, I=8960, Hq=12, Hkv=2, D=128, w4=False),
        "7b": dict(H=3584, I=18944, Hq=28, Hkv=4, D=128, w4=False),
        "mistral": dict(H=4096, I=14336, Hq=32, Hkv=8, D=128, w4=False),
        "phi-w4": dict(H=3072, I=8192, Hq=32, Hkv=32, D=96, w4=True)}


def copies(nbytes):
    return max(2, min(24, int(1.2e9 // nbytes) + 1))


def bf(*shape, scale=0.05):
    return torch.empty(*shape, dtype=BF, device="cuda").normal_(0, scale)


def w4(N, K):
    wq = torch.randint(-2 ** 31, 2 ** 31 - 1, (N, K // 8), device="cuda", dtype=torch.int64).to(torch.int32)
    sc = (0.005 * (0.75 + 0.5 * torch.rand(N, K // 64, device="cuda"))).to(BF)
    bi = (sc.float() * -7.5).to(BF)
    bits = lambda t: t.view(torch.int16).to(torch.int64) & 0xFFFF  # noqa: E731
    sb = bits(sc) | (bits(bi) << 16)
    sb = torch.where(sb >= 2 ** 31, sb - 2 ** 32, sb).to(torch.int32)
    return wq.contiguous(), sb.contiguous()


def timed(fn, n_copies, reps):
    """us per launch: `reps` launches captured in ONE graph (no host launch cost between them), best of 3 replays"""
    for i in range(n_copies):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for r in range(reps):
            fn(r % n_copies)
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def run(name, M, reps):
    d = DIMS[name]
    H, I, Hq, Hkv, D, q = d["H"], d["I"], d["Hq"], d["Hkv"], d["D"], d["w4"]
    x, xi = bf(M, H, scale=1.0), bf(M, I, scale=1.0)
    nw = (1 + 0.1 * torch.randn(H, device="cuda")).to(BF)
    out = []

    def add(tag, N, K, fn_of_w, mk):
        nbytes = N * K // 2 + N * K // 16 if q else N * K * 2
        n = copies(nbytes)
        ws = [mk(N, K) for _ in range(n)]
        us = timed(lambda i: fn_of_w(ws[i]), n, reps)
        out.append((tag, N, K, nbytes, us))
        del ws
        torch.cuda.empty_cache()

    mk = (lambda N, K: w4(N, K)) if q else (lambda N, K: bf(N, K))
    gv = (lambda x_, w, **kw: vops.gemv_w4_ws(x_, w[0], w[1], **kw)) if q else (lambda x_, w, **kw: vops.gemv_ws(x_, w, **kw))
    # qkv: RMSNorm + bias (the RoPE + KV-write epilogue form needs the paged pools: timed below when M >= 9 and bf16)
    Nqkv = (Hq + 2 * Hkv) * D
    bq = bf(Nqkv, scale=0.3)
    add("qkv norm+bias", Nqkv, H, lambda w: gv(x, w, norm_w=nw, bias=bq, epilogue=vops.EPI_BIAS), mk)
    if M >= 9 and not q and D == 128:
        max_pages = 8
        pos = torch.arange(M, dtype=torch.int32, device="cuda") + 400
        slot = torch.full((M,), 70, dtype=torch.int32, device="cuda")
        inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))).cuda()
        kpool = torch.zeros(M * max_pages, Hkv, D // 8, 64, 8, dtype=BF, device="cuda")
        vpool = torch.zeros(M * max_pages, Hkv, D, 64, dtype=BF, device="cuda")
        qkv_out = torch.zeros(M, Nqkv, dtype=BF, device="cuda")
        add("qkv norm+rope+kv", Nqkv, H,
            lambda w: vops.gemv_qkv_rope_kvwrite_ws(x, nw, w, bq, Hq, Hkv, D, pos, slot, inv, None, kpool, vpool, out=qkv_out,
                                                    max_pages=max_pages), mk)
    res = bf(M, H, scale=1.0)
    xo = bf(M, Hq * D, scale=1.0)
    add("o_proj +res", H, Hq * D, lambda w: gv(xo, w, res=res, out=res, epilogue=vops.EPI_RESIDUAL), mk)
    add("gate/up norm+swiglu", 2 * I, H, lambda w: gv(x, w, norm_w=nw, epilogue=vops.EPI_SWIGLU), mk)
    add("down +res", H, I, lambda w: gv(xi, w, res=res, out=res, epilogue=vops.EPI_RESIDUAL), mk)
    tot_us = sum(o[4] for o in out if o[0] != "qkv norm+bias" or len(out) == 4)
    tot_b = sum(o[3] for o in out if o[0] != "qkv norm+bias" or len(out) == 4)
    print(f"== {name} rows={M}")
    for tag, N, K, nbytes, us in out:
        print(f"  {tag:22s} N={N:6d} K={K:6d} {nbytes / 1e6:8.1f} MB {us:8.2f} us {nbytes / us / 1e6:6.2f} TB/s")
    print(f"  layer (projections)    {tot_b / 1e6:8.1f} MB {tot_us:8.2f} us {tot_b / tot_us / 1e6:6.2f} TB/s", flush=True)


def run_gemm(name, M, reps):
    """the same projections through the prefill GEMMs (vlm_gemm_bf16): what a decode step of MORE than 16 rows would use"""
    d = DIMS[name]
    H, I, Hq, Hkv, D = d["H"], d["I"], d["Hq"], d["Hkv"], d["D"]
    if d["w4"]:
        return
    x, xi, xo = bf(M, H, scale=1.0), bf(M, I, scale=1.0), bf(M, Hq * D, scale=1.0)
    res = bf(M, H, scale=1.0)
    Nqkv = (Hq + 2 * Hkv) * D
    bq = bf(Nqkv, scale=0.3)
    out = []

    def add(tag, N, K, fn):
        nbytes = N * K * 2
        n = copies(nbytes)
        ws = [bf(N, K) for _ in range(n)]
        us = timed(lambda i: fn(ws[i]), n, reps)
        out.append((tag, N, K, nbytes, us))
        del ws
        torch.cuda.empty_cache()

    oq = torch.empty(M, Nqkv, dtype=BF, device="cuda")
    oa = torch.empty(M, I, dtype=BF, device="cuda")
    add("qkv +bias (gemm)", Nqkv, H, lambda w: vops.gemm(x, w, bias=bq, out=oq, epilogue=vops.EPI_BIAS))
    add("o_proj +res (gemm)", H, Hq * D, lambda w: vops.gemm(xo, w, res=res, out=res, epilogue=vops.EPI_RESIDUAL))
    add("gate/up swiglu (gemm)", 2 * I, H, lambda w: vops.gemm(x, w, out=oa, epilogue=vops.EPI_SWIGLU))
    add("down +res (gemm)", H, I, lambda w: vops.gemm(xi, w, res=res, out=res, epilogue=vops.EPI_RESIDUAL))
    tot_us, tot_b = sum(o[4] for o in out), sum(o[3] for o in out)
    print(f"== {name} rows={M} through the prefill GEMMs")
    for tag, N, K, nbytes, us in out:
        print(f"  {tag:22s} N={N:6d} K={K:6d} {nbytes / 1e6:8.1f} MB {us:8.2f} us {nbytes / us / 1e6:6.2f} TB/s")
    print(f"  layer (projections)    {tot_b / 1e6:8.1f} MB {tot_us:8.2f} us {tot_b / tot_us / 1e6:6.2f} TB/s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gemm-rows", default="")
    ap.add_argument("models", nargs="*", default=["2b", "7b"])
    ap.add_argument("--rows", default="16")
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    knobs = {k: v for k, v in os.environ.items() if k.startswith("VLM_GEMV_MFMA")}
    print("knobs:", knobs or "defaults")
    for name in a.models:
        for M in [int(r) for r in a.rows.split(",") if r]:
            run(name, M, a.reps)
        for M in [int(r) for r in a.gemm_rows.split(",") if r]:
            run_gemm(name, M, a.reps)


if __name__ == "__main__":
    main()
