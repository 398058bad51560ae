#!/bin/bash
# skinny-M MFMA decode GEMM: parity tests, then batch-8 / batch-1 A-B in bench-like runs
timeout 500 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_parity_decode_gpu.py -x -q -m gpu -k "gemv or batch or teacher or peaked" 2>&1 | tail -8
cat > /tmp/ab.py <<'P'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl
dev = torch.device("cuda", 0)
cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=16384, max_seqs=16)
lm = model.language_model
for on in (1, 0, 1, 0):
    lm.apply_tuning(mfma_gemv=on)
    r = bench.batch_decode_throughput(model, cfg, 8, 64)
    req = bench.build_request(cfg, 448, 128, seed=0); req = (req[0], req[1].to(dev), req[2])
    bench.run_step(model, req, 64, 8)
    a, b, toks = bench.run_step(model, req, 256, 8)
    print(f"mfma_gemv={on}: batch8 {r['generation_tps']:.0f} tok/s, batch1 {255 / b:.1f} tok/s", flush=True)
P
timeout 300 python /tmp/ab.py 2>&1 | tail -5


