"""A/B of the skinny-M MFMA decode GEMM against the v_dot2c GEMVs inside one process (batch 8 and batch 1)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl
dev = torch.device("cuda", 0)
cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=16384, max_seqs=16)
lm = model.language_model
for on in (1, 0, 1, 0):
    lm.apply_tuning(mfma_gemv=on)
    r = bench.batch_decode_throughput(model, cfg, 8, 64)
    r4 = bench.batch_decode_throughput(model, cfg, 4, 64)
    print(f"mfma_gemv={on}: batch8 {r['generation_tps']:.0f} tok/s, batch4 {r4['generation_tps']:.0f} tok/s", flush=True)
