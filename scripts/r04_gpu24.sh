#!/bin/bash
# Session 24 (LDS histogram, coalesced rank / radix passes): the sampler surface (vlm_sample_ex) against the reference's golden rows and the oracle, the older sampler /
# eager-step tests around it, and what a typical-p call costs at V = 151,936.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu24
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_sampler_gpu.py -q -m gpu --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -80 ) > $O/t_sampler.log 2>&1; tail -40 $O/t_sampler.log
( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_parity_decode_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "sampl or python or processor or penalt or abi or symbols" 2>&1 | tail -15 ) > $O/t_old.log 2>&1; tail -8 $O/t_old.log
timeout 120 python - > $O/timing.log 2>&1 <<'P'
import sys, torch
sys.path.insert(0, '.')
from mlx_vlm_amd import ops
V = 151936
x = (torch.randn(1, V) * 2).to(torch.bfloat16).cuda()
step = torch.zeros(1, dtype=torch.int32, device='cuda')
for name, kw in (("top_p", dict(top_p=0.9)), ("classic chain", dict(top_p=0.9, min_p=0.02, top_k=50)), ("top_n_sigma", dict(top_n_sigma=1.0)),
                 ("p_less", dict(p_less=True)), ("typical_p", dict(typical_p=0.9)), ("xtc", dict(xtc_probability=1.0, xtc_threshold=0.01)),
                 ("min_keep", dict(min_p=0.5, min_tokens_to_keep=100))):
    for _ in range(3):
        ops.sample(x, temperature=0.8, step=step, want_logprobs=False, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.sample(x, temperature=0.8, step=step, want_logprobs=False, **kw)
    b.record(); torch.cuda.synchronize()
    print(f"{name:16s} {a.elapsed_time(b) * 100:.1f} us per vlm_sample_ex call (V = {V}, 1 row, 4-5 launches + allocations)")
P
cat $O/timing.log
