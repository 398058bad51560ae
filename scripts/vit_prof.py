#!/usr/bin/env python
"""ViT-only run for profiling: 16 x 336^2 images per call (9216 patches), Qwen2-VL-2B vision tower."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models.qwen2_vl import ModelConfig
from mlx_vlm_amd.models.qwen2_vl.vision import VisionModel
cfg = ModelConfig.from_dict(dict(synthetic.QWEN2_VL_2B))
W = synthetic.random_weights(cfg, seed=0, device="cuda")
vt = VisionModel(cfg.vision_config)
vt.load_weights({k[len("vision_tower."):]: v for k, v in W.items() if k.startswith("vision_tower.")})
del W
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
side = int(sys.argv[2]) // 14 if len(sys.argv) > 2 else 24      # argv[2]: image side in pixels (336 -> 24 x 24 patches, 448 -> 32 x 32)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
pix = torch.randn(nimg * side * side, 1176, device="cuda")
thw = np.array([[1, side, side]] * nimg)
for _ in range(2): vt(pix, thw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): vt(pix, thw)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print(f"{nimg} images: {dt*1e3:.2f} ms/call, {nimg/dt:.1f} img/s, {nimg/dt*0.791:.1f} TFLOP/s")
