"""ViT tower throughput vs images per call (336 x 336), Qwen2-VL-2B dims: how much of the 0.35 is tile quantisation at 16 images"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=4096, max_seqs=4)
out = {}
for n in [int(a) for a in (sys.argv[1:] or ["8", "16", "24", "32", "48", "64", "96", "128"])]:
    ips, dt = bench.vit_throughput(model, cfg, n, 336)
    out[n] = {"images_per_s": ips, "ms_per_call": dt * 1e3, "tflops": ips * bench.VIT_TFLOP_336, "frac": ips * bench.VIT_TFLOP_336 / 2500.0}
    print(n, out[n], flush=True)
print(json.dumps(out))
