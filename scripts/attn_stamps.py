import os, sys
os.environ["VLM_ATTN_STAMPS"] = "1"   # needs a library built with -DVLM_ATTN_TIMELINE (debug timeline stamps)
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models.qwen2_vl import Model, ModelConfig
from mlx_vlm_amd.generate import generate_step
cfg = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))
W = synthetic.random_weights(cfg, seed=0, device="cuda")
model = Model(cfg, kv_pool_tokens=8192, max_seqs=8); model.load_weights(W); del W
ids = np.random.default_rng(0).integers(0, 150000, (1, 600))
n = 0
for tok, _ in generate_step(ids, model, None, None, max_tokens=20, return_logprobs=False, use_graph=False):
    n += 1
torch.cuda.synchronize()
st = model.language_model.decode_state(1)
names = ["start", "q issued", "kv issued", "QK done", "softmax done", "PV done", "pre-barrier", "post-barrier", "end"]
po = st.part_o.reshape(-1)[:32].cpu().tolist()
for w, off in (("wave0", 0), ("wave15", 16)):
    print(w, " ".join(f"{names[i]}={po[off+i]:.2f}" for i in range(9)))
