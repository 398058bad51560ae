"""batch-8 / batch-16 decode throughput of the default workload's model (bench.batch_decode_throughput), three runs each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl
dev = torch.device("cuda", 0)
cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=32768, max_seqs=40)
for B in (8, 16):
    print(f"batch {B}:", " ".join(f"{bench.batch_decode_throughput(model, cfg, B, 64)['generation_tps']:.0f}" for _ in range(3)), "tok/s", flush=True)
