"""nanoLLaVA (BASELINE configs[0]) on one MI355X: decode tokens/s, time to first token and SigLIP
tower images/s at the real dims (Qwen1.5-0.5B + SigLIP-so400m/14-384, random-init bf16), in the format of bench.py.
Not a driver contract file: the headline stays `bench.py` (Qwen2-VL-2B).

    python scripts/bench_nanollava.py [--steps 3] [--warmup 1] [--max-tokens 64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mlx_vlm_amd import synthetic  # noqa: E402
from mlx_vlm_amd.models.llava_bunny import ImageProcessor, Model, ModelConfig  # noqa: E402
from mlx_vlm_amd.utils import freeze_heap  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--max-tokens", type=int, default=64)        # BASELINE configs[0]: 64 new tokens
    ap.add_argument("--lookahead", type=int, default=8)
    args = ap.parse_args()
    cfg = ModelConfig.from_dict(dict(synthetic.NANOLLAVA))
    W = synthetic.random_weights(cfg, seed=0, device="cuda")
    model = Model(cfg, device="cuda", kv_pool_tokens=8192, max_seqs=8)
    model.load_weights(W)
    del W
    freeze_heap()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (336, 336, 3), dtype=np.uint8)                  # north_star image size; resized to 384
    pix = torch.from_numpy(np.stack(ImageProcessor().preprocess([img]))).cuda()
    text = np.random.default_rng(1000).integers(0, 151643, 128)
    ids = np.concatenate([text[:64], [cfg.image_token_index], text[64:]]).astype(np.int64)[None]
    req = (ids, pix, None)

    def step():
        from mlx_vlm_amd.generate import generate_step

        t0 = time.perf_counter()
        toks, t_first = [], None
        for tok, _ in generate_step(ids, model, pix, None, max_tokens=args.max_tokens, temperature=0.0,
                                    return_logprobs=False, lookahead=args.lookahead):
            if t_first is None:
                t_first = time.perf_counter()
            toks.append(tok)
        torch.cuda.synchronize()
        return t_first - t0, time.perf_counter() - t_first, toks

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    pre, dec = 0.0, 0.0
    for _ in range(args.steps):
        a, b, toks = step()
        pre, dec = pre + a, dec + b
    n_dec = args.steps * (args.max_tokens - 1)
    t, v = cfg.text_config, cfg.vision_config
    per_layer = 4 * t.hidden_size * t.hidden_size + 3 * t.hidden_size * t.intermediate_size
    bytes_per_token = 2 * (t.num_hidden_layers * per_layer + t.vocab_size * t.hidden_size)     # real (unpadded) weights
    prompt_tokens = ids.shape[1] - 1 + model.vision_tower.num_patches

    def tower(n):
        batch = pix.expand(n, -1, -1, -1).contiguous()
        for _ in range(2):
            model.vision_tower(batch)
        torch.cuda.synchronize()
        dts = sorted(bench.time_events(lambda: model.vision_tower(batch), 1) for _ in range(5))
        return n / dts[2]

    N, E, I = model.vision_tower.num_patches, v.hidden_size, v.intermediate_size
    tflop = (v.num_hidden_layers * (2 * N * E * 3 * E + 2 * N * E * E + 4 * N * E * I + 4 * N * N * E)
             + 2 * N * model.vision_tower.patch_dim * E) / 1e12
    ips1, ips8 = tower(1), tower(8)
    tps = n_dec / dec
    out = {"metric": "decode tokens/sec + vision-prefill images/sec, nanoLLaVA", "value": tps, "unit": "tokens/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "nanoLLaVA dims (Qwen1.5-0.5B + SigLIP-so400m/14-384, random-init bf16), one 336x336 image "
                                  "resized to 384x384 (729 image tokens) + 128 text tokens, greedy decode, EOS disabled",
                      "prompt_tokens": int(prompt_tokens), "max_tokens": args.max_tokens},
           "decode_us_per_token": 1e6 / tps, "prefill_ms_to_first_token": pre / args.steps * 1e3,
           "prompt_tps": prompt_tokens * args.steps / pre,
           "roofline_decode_step": {"bound": "hbm", "achieved": bytes_per_token * tps / 1e9, "peak": bench.HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": bytes_per_token * tps / 1e9 / bench.HBM_PEAK_GBS,
                                    "traffic": None, "algorithmic_bytes_per_token": bytes_per_token,
                                    "launches_per_token": int(model.language_model.decode_launches())
                                    if hasattr(model.language_model, "decode_launches") else None},
           "vision_images_per_s": {"1_per_call": ips1, "8_per_call": ips8},
           "roofline_vit": {"bound": "mfma", "achieved": ips8 * tflop, "peak": bench.MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                            "frac": ips8 * tflop / bench.MFMA_BF16_PEAK_TF, "tflop_per_image": tflop, "traffic": None}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
