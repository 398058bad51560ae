#!/usr/bin/env python
"""How much of the decode time per token is host-side pipeline (token read-back, events) and how much is the replayed
graph itself?  Replays the captured decode step N times back to back with no read-back and compares with generate_step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mlx_vlm_amd import ops, synthetic
from mlx_vlm_amd.models import cache as cache_mod
from mlx_vlm_amd.models.qwen2_vl import Model, ModelConfig
from mlx_vlm_amd.sample_utils import make_sampler

cfg = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))
W = synthetic.random_weights(cfg, seed=0, device="cuda")
model = Model(cfg, kv_pool_tokens=16384, max_seqs=16); model.load_weights(W); del W
lm = model.language_model
ids, pix, thw = bench.build_request(cfg, 448, 128, 0)
for lookahead in (8,):
    a, b, toks = bench.run_step(model, (ids, pix.cuda(), thw), 256, lookahead)
    a, b, toks = bench.run_step(model, (ids, pix.cuda(), thw), 256, lookahead)
    print(f"generate_step lookahead {lookahead}: {255 / b:.1f} tok/s ({b / 255 * 1e6:.1f} us per token)")
f = model.get_input_embeddings(ids, pix.cuda(), image_grid_thw=thw)
pc = cache_mod.make_prompt_cache(lm)
emb = f.inputs_embeds; L = emb.shape[1]
pos = np.asarray(f.position_ids)
logits = lm.prefill(emb.reshape(L, -1), pos.reshape(3, L), [pc], [L], "last", reserve_extra=600)
sargs = make_sampler(temp=0.0).engine_args()
tok0, _ = ops.sample(logits, step=torch.zeros(1, dtype=torch.int32, device="cuda"), want_logprobs=False, **sargs)
st = lm.decode_begin([pc], tok0, np.asarray(f.rope_deltas).reshape(-1)[:1], max_new_tokens=600)
st.step.fill_(1)
G = 1   # (several steps per captured graph were tried: 1 / 4 / 16 steps per graph = 988 / 985 / 955 tok/s - no inter-graph gap to win)
lm.decode_run(st, 8, sargs, use_graph=True)
torch.cuda.synchronize()
for n in (256 // G,):
    t0 = time.perf_counter()
    lm.decode_run(st, n, sargs, use_graph=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"bare graph replay x{n} ({G} steps per graph): {n * G / dt:.1f} tok/s ({dt / (n * G) * 1e6:.1f} us per token)")
