"""Why did bench.py's continuous-batching extra read 885 useful tok/s when the same queue runs at ~2000 in
scripts/continuous_ab.py?  Replays bench.py's order of extras in one process and times every BatchGenerator.next(),
printing the slow rounds with what the allocator / GC did meanwhile."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mlx_vlm_amd import synthetic  # noqa: E402
from mlx_vlm_amd.batch import BatchGenerator  # noqa: E402
from mlx_vlm_amd.models import qwen2_vl  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=16384, max_seqs=16)
    gcs = []
    t0 = {}

    def cb(phase, info):
        if phase == "start":
            t0[info["generation"]] = time.perf_counter()
        else:
            gcs.append((time.perf_counter(), info["generation"], 1e3 * (time.perf_counter() - t0.get(info["generation"], 0))))

    gc.callbacks.append(cb)
    if "--pre" in sys.argv:
        req = bench.build_request(cfg, 448, 128, seed=0)
        req = (req[0], req[1].to(dev), req[2])
        for _ in range(2):
            bench.run_step(model, req, 256, 8)
        bench.kernel_rooflines(model, cfg)
        bench.vit_throughput(model, cfg, 16, 336)
        bench.vit_throughput(model, cfg, 1, 448)
        bench.batch_decode_throughput(model, cfg, 8, 64)
    n_requests, rows = 24, 8
    reqs = [bench.build_request(cfg, 336, 64, 700 + i) for i in range(n_requests)]
    ids = [r[0].reshape(-1) for r in reqs]
    kw = [dict(pixel_values=r[1], image_grid_thw=r[2]) for r in reqs]
    lens = [24 + (37 * i) % 73 for i in range(n_requests)]
    parts = {}

    def timed(name):
        fn = getattr(BatchGenerator, name)

        def w(self, *a, **k):
            t = time.perf_counter()
            try:
                return fn(self, *a, **k)
            finally:
                parts[name] = parts.get(name, 0.0) + time.perf_counter() - t
        setattr(BatchGenerator, name, w)

    from mlx_vlm_amd import _lib
    ring_t = {"wait": 0.0, "memcpy": 0.0, "enqueue": 0.0, "n": 0, "max_memcpy": 0.0, "max_enqueue": 0.0}

    def stage(self, t, device, out=None):
        n = t.numel() * t.element_size()
        if n == 0 or n > self.buf.numel() // 2:
            return t.to(device) if out is None else out.copy_(t)
        start = (self.head + 255) & ~255
        if start + n > self.buf.numel():
            start = 0
        end = start + n
        a = time.perf_counter()
        newest = -1
        for i, (s0, e0, _) in enumerate(self.inflight):
            if s0 < end and start < e0:
                newest = i
        if newest >= 0:
            self.inflight[newest][2].synchronize()
            for _ in range(newest + 1):
                self.inflight.popleft()
        while self.inflight and self.inflight[0][2].query():
            self.inflight.popleft()
        b = time.perf_counter()
        view = self.buf[start:end].view(t.dtype).view(t.shape)
        if "--torch-copy" in sys.argv:
            view.copy_(t)                              # the old form: torch's 128-thread CPU copy
        else:
            import ctypes
            ctypes.memmove(self.buf.data_ptr() + start, t.contiguous().data_ptr(), n)
        c = time.perf_counter()
        out = view.to(device, non_blocking=True) if out is None else out.copy_(view, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        d = time.perf_counter()
        ring_t["wait"] += b - a; ring_t["memcpy"] += c - b; ring_t["enqueue"] += d - c; ring_t["n"] += 1
        ring_t["max_memcpy"] = max(ring_t["max_memcpy"], c - b); ring_t["max_enqueue"] = max(ring_t["max_enqueue"], d - c)
        self.inflight.append((start, end, ev))
        self.head = end
        return out

    _lib._PinnedRing.stage = stage
    for name in ("_admit_begin", "_admit_join", "_launch_step", "_prefill_requests", "_drop_rows"):
        timed(name)
    for rep in range(4):
        parts.clear()
        for k in ring_t:
            ring_t[k] = 0
        ms0 = torch.cuda.memory_stats()
        gen = BatchGenerator(model, None, completion_batch_size=rows, prefill_batch_size=rows, compute_logprobs=False)
        torch.cuda.synchronize()
        prof = None
        if rep == 3 and "--cprofile" in sys.argv:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        w0 = time.perf_counter()
        gen.insert(ids, lens, prompt_kwargs=kw)
        rounds = []
        n = 0
        while gen.has_work:
            a = time.perf_counter()
            n += len(gen.next()[1])
            rounds.append((a, time.perf_counter() - a))
        gen.close()
        torch.cuda.synchronize()
        wall = time.perf_counter() - w0
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof).sort_stats("tottime").print_stats(22)
        ms1 = torch.cuda.memory_stats()
        slow = [(i, 1e3 * d) for i, (_, d) in enumerate(rounds) if d > 0.01]
        print(f"rep{rep}: {n / wall:7.1f} useful tok/s, {len(rounds)} rounds, wall {1e3 * wall:.1f} ms, sum(next) {1e3 * sum(d for _, d in rounds):.1f} ms; "
              f"slow rounds (>10 ms): {[(i, round(d, 1)) for i, d in slow]}")
        print("      host ms by part: " + ", ".join(f"{k} {1e3 * v:.1f}" for k, v in parts.items()))
        print("      pinned ring ms: " + ", ".join(f"{k} {v if k == 'n' else round(1e3 * v, 2)}" for k, v in ring_t.items()))
        print(f"      device mallocs +{ms1['num_device_alloc'] - ms0['num_device_alloc']}, frees +{ms1['num_device_free'] - ms0['num_device_free']}, "
              f"retries +{ms1['num_alloc_retries'] - ms0['num_alloc_retries']}, reserved {ms1['reserved_bytes.all.current'] / 2**30:.2f} GiB; "
              f"gc passes in window: {[(g, round(ms, 1)) for (t, g, ms) in gcs if t >= w0]}")


if __name__ == "__main__":
    main()
