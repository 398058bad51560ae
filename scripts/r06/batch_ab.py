#!/usr/bin/env python
"""batched decode of the 2B engine (bench.py's batch_decode_throughput) - one process per setting of the A/B knob in the environment
usage: batch_ab.py [rows ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl
dev = torch.device("cuda", 0)
cfg, model, load = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=32768, max_seqs=40)
for B in [int(a) for a in sys.argv[1:]] or [16, 8]:
    r = bench.batch_decode_throughput(model, cfg, B, 64)
    print(f"rows {B}: " + json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (dict, list))}), flush=True)
