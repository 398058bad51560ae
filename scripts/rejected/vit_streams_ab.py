"""A/B of the image-parallel ViT launch chains (vlm_vit_forward_parts): 16 / 32 x 336x336 images per call at Qwen2-VL-2B
tower dims, VLM_VIT_STREAMS = 1 (one chain, the round-3 form) .. 4, median of 9 single-call HIP-event timings each,
interleaved (the variants see the same clocks).  Output: ms per call, TFLOP/s, fraction of 2.5 PF."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mlx_vlm_amd import synthetic  # noqa: E402
from mlx_vlm_amd.models import qwen2_vl  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=4096, max_seqs=4)
    for n_images in (16, 32, 8):
        reqs = [bench.build_request(cfg, 336, 1, 100 + i) for i in range(n_images)]
        pix = torch.cat([r[1] for r in reqs], dim=0).cuda()
        thw = np.concatenate([r[2] for r in reqs], axis=0)
        variants = ["1", "2", "3", "4"]
        ref = None
        for v in variants:
            os.environ["VLM_VIT_STREAMS"] = v
            out = model.vision_tower(pix, thw)
            out = model.vision_tower(pix, thw)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                same = torch.equal(ref, out)
                d = (ref.float() - out.float()).abs().max().item() / (ref.float().abs().max().item() + 1e-30)
                print(f"{n_images:3d} images, {v} chains vs 1: {'bit-identical' if same else 'max |diff| / max |ref| = %.2e (split-K choice of the merger GEMMs depends on the rows per call)' % d}", flush=True)
        times = {v: [] for v in variants}
        for _ in range(9):
            for v in variants:
                os.environ["VLM_VIT_STREAMS"] = v
                times[v].append(bench.time_events(lambda: model.vision_tower(pix, thw), 1))
        for v in variants:
            dt = sorted(times[v])[len(times[v]) // 2]
            tf = n_images * bench.VIT_TFLOP_336 / dt
            print(f"{n_images:3d} images, {v} chain(s): {dt * 1e3:8.3f} ms per call  {n_images / dt:8.1f} img/s  {tf:7.1f} TFLOP/s  "
                  f"{tf / bench.MFMA_BF16_PEAK_TF:.4f} of peak   (min {min(times[v]) * 1e3:.3f} ms)", flush=True)


if __name__ == "__main__":
    main()
