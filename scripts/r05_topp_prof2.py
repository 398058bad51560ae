"""the sampled top-p step from LOGITS (the captured decode step's form), 30 calls, for a rocprofv3 kernel trace; VLM_SAMPLE_SPLIT=0 in
the environment gives the one-workgroup route for the same table"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_vlm_amd import ops
torch.manual_seed(0)
x = (torch.randn(1, 151936, device="cuda") * 2).to(torch.bfloat16)
ws = ops.sample_workspace(1, "cuda")
st = torch.zeros(1, dtype=torch.int32, device="cuda")
for _ in range(30):
    ops.sample(x, temperature=0.7, seed=1, step=st, want_logprobs=True, ws=ws, top_p=0.9)
torch.cuda.synchronize()
