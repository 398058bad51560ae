"""Where the end-to-end time of the Qwen2-VL-7B batch-32 job goes on the host: cProfile around ONE dp_batch_generate call of
bench.py's `qwen2vl-7b-b32` workload (same model, same requests), plus wall / decode / prompt seconds of that call.
    python scripts/profile_7b_host.py [n_requests] [max_tokens]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    max_tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.models import qwen2_vl

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_7B, qwen2_vl, 0, dev, kv_pool_tokens=32768, max_seqs=40)
    reqs = []
    for i in range(n_req):
        ids, pix, thw = bench.build_request(cfg, 336, 128, seed=i)
        reqs.append({"input_ids": ids.reshape(-1), "pixel_values": pix, "image_grid_thw": thw, "max_tokens": max_tokens})
    parallel.dp_batch_generate(model, None, requests=reqs[:2], max_tokens=8)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = parallel.dp_batch_generate(model, None, requests=reqs, max_tokens=max_tokens)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(f"rep {rep}: wall {wall * 1e3:.1f} ms, decode {res['decode_time_s'] * 1e3:.1f} ms in {res['decode_steps']} steps, "
              f"{res['generation_tokens']} tokens -> {res['generation_tokens'] / wall:.0f} tok/s end to end; keys {sorted(res)}")
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    parallel.dp_batch_generate(model, None, requests=reqs, max_tokens=max_tokens)
    torch.cuda.synchronize()
    pr.disable()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32)
        print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
