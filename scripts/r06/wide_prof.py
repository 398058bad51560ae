"""The 2B engine at `rows` (argv[1], default 64) concurrent requests through wide steps - for rocprofv3 / A-B runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from benchlib.extras import wide_decode_throughput
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models.qwen2_vl import ModelConfig
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))
print(json.dumps(wide_decode_throughput(None, cfg, rows, 48)))
