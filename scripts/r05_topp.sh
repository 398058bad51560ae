#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_topp; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_sampler_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "sampler or split or filters or sample or draw or xtc or typical" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
for sp in 0 1; do
  VLM_SAMPLE_SPLIT=$sp timeout 300 python3 - > $O/time$sp.out 2>&1 <<'P'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from mlx_vlm_amd import ops
torch.manual_seed(0)
x = torch.randn(1, 151936, device="cuda") * 2
lp = (x - torch.logsumexp(x, -1, keepdim=True)).to(torch.bfloat16)
ws = ops.sample_workspace(1, "cuda")
st = torch.zeros(1, dtype=torch.int32, device="cuda")
def run(**kw): return ops.sample(lp, temperature=0.7, seed=1, step=st, want_logprobs=False, input_is_logprobs=True, ws=ws, **kw)
for name, kw in (("plain", {}), ("top_p", dict(top_p=0.9)), ("top_k", dict(top_k=50)), ("chain", dict(top_p=0.9, min_p=0.02, top_k=50))):
    for _ in range(5): run(**kw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(**kw); s.synchronize()
        with torch.cuda.graph(g, stream=s): run(**kw)
    torch.cuda.current_stream().wait_stream(s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize(); a.record()
    for _ in range(50): g.replay()
    b.record(); b.synchronize()
    print(f"split={os.environ.get('VLM_SAMPLE_SPLIT')} {name:6s}: {a.elapsed_time(b) * 1e3 / 50:7.1f} us per sampler call (graph replay)")
P
done
timeout 600 python3 bench.py --stage extras --gpus 1 --steps 2 --warmup 1 --max-tokens 256 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sampled', d.get('sampled'))" > $O/sampled.out 2>&1
cat $O/rc.txt; tail -3 $O/pytest.out; cat $O/time0.out $O/time1.out | grep split; cat $O/sampled.out
