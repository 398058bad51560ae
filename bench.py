"""Headline benchmark: decode tokens/sec + vision-prefill images/sec, Qwen2-VL-2B bf16 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload qwen2vl-2b | nanollava | qwen2vl-7b-b32 | qwen2vl-2b-w4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`): the default is configs[1] (below); `nanollava` is configs[0] (nanoLLaVA dims, one
336x336 image resized to 384x384, greedy 64 tokens); `qwen2vl-7b-b32` is configs[2] (Qwen2-VL-7B dims, 32 requests
dealt data-parallel over the ranks through parallel.dp_batch_generate, every rank a continuous BatchGenerator).

A "step" = one pass of the hot path over one request of BASELINE.json configs[1]:
one synthetic 448x448 image (1024 patches -> 256 image tokens) + 128 text tokens,
ViT prefill -> projector -> LLM prefill -> 256 greedy tokens (EOS disabled), batch 1
per GPU.  N > 1: one process per GPU (RCCL), weights broadcast from rank 0 at load,
every rank serves its own request stream (weak scaling, no collective in the step).
Rank 0 prints ONE JSON line; `value` = whole-job decode tokens/s.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the parts (benchlib/): request + loading + roofline constants, the timed region, the extras, the other configs, the CPU baselines.
# Re-exported here: scripts/ and tests/ use this file as one module.
from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.cpu_baseline import *  # noqa: E402,F401,F403
from benchlib.extras import *  # noqa: E402,F401,F403
from benchlib.headline import *  # noqa: E402,F401,F403
from benchlib.workloads import *  # noqa: E402,F401,F403


def dry_run(args, rank, ws):
    """Everything of the N-rank job EXCEPT the kernels, on CPU ranks over gloo: the self-launch / torchrun environment, the
    rendezvous, the weight replica through parallel.WeightArena (rank 0 fills it, the others receive in place; contents
    verified on every rank), dp_batch_generate's length-sorted request deal + gather with a mock per-rank engine, and the
    timing protocol of the real line (warm-up, barrier, K steps, barrier, MAX over ranks).  The 8-GPU scaling run is the
    driver's; this keeps a launcher or collective typo from costing that run."""
    from mlx_vlm_amd import parallel

    dev = torch.device("cpu")
    t0 = time.perf_counter()
    W = {f"layers.{i}.w": (torch.full((1 << 20,), float(i + 1), dtype=torch.bfloat16) if rank == 0 else torch.empty(1 << 20, dtype=torch.bfloat16))
         for i in range(8)}
    W["table"] = torch.arange(4096, dtype=torch.int32) if rank == 0 else torch.empty(4096, dtype=torch.int32)
    parallel.barrier()
    t1 = time.perf_counter()
    parallel.broadcast_weights(W, src=0, bucket_bytes=4 << 20)
    bcast_s = parallel.max_over_ranks(time.perf_counter() - t1, dev)
    for i in range(8):
        assert float(W[f"layers.{i}.w"].float().mean()) == float(i + 1), ("broadcast", rank, i)
    assert W["table"][-1].item() == 4095
    nbytes = sum(v.numel() * v.element_size() for v in W.values())
    load = {"load_s": time.perf_counter() - t0, "weight_bytes": nbytes, "broadcast_s": bcast_s}
    max_tokens = args.max_tokens or 8

    def serve(indices, reqs, max_toks):                    # the per-rank engine: token j of request i = (i * 31 + j) % 997
        time.sleep(0.001 * len(indices))
        return [[(i * 31 + j) % 997 for j in range(max_toks[i])] for i in indices]

    rng = np.random.default_rng(0)                          # the same list on every rank
    reqs = [{"input_ids": np.arange(5 + int(rng.integers(0, 50)))} for _ in range(4 * ws + 1)]
    for _ in range(args.warmup):
        parallel.dp_batch_generate(None, None, requests=reqs, max_tokens=max_tokens, serve=serve)
    parallel.barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = parallel.dp_batch_generate(None, None, requests=reqs, max_tokens=max_tokens, serve=serve)
    parallel.barrier()
    my_wall = time.perf_counter() - t0
    wall = parallel.max_over_ranks(my_wall, dev)
    # the same per-rank vectors the real line carries (stage_headline), through the same collective
    decode_ranks = parallel.per_rank(my_wall, dev)
    thread_ranks = parallel.per_rank(float(torch.get_num_threads()), dev)
    if rank != 0:
        return None
    assert res["tokens"] == [[(i * 31 + j) % 997 for j in range(max_tokens)] for i in range(len(reqs))]
    info = _dist_info(ws, load)
    return {"metric": "decode tokens/sec + vision-prefill images/sec, Qwen2-VL-2B", "dry_run": True,
            "value": args.steps * res["generation_tokens"] / wall, "unit": "tokens/s", "n_gpus": ws, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "none (no kernels)", "data": "synthetic",
            "config": {"workload": "DRY RUN: mock per-rank engine over gloo CPU ranks - launcher / rendezvous / weight arena "
                                   "broadcast / request deal / timing protocol only", "parallelism": f"dp{ws}"},
            "load": load, "distributed": info, "per_rank_requests": res["per_rank_requests"],
            "host_prep_s_per_rank": res["host_prep_s_per_rank"], "serve_s_per_rank": res["serve_s_per_rank"],
            "decode_s_per_rank": decode_ranks, "torch_threads_per_rank": [int(v) for v in thread_ranks]}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--max-tokens", type=int, default=0, help="0 = the workload's own (256 / 64 / 64)")
    ap.add_argument("--lookahead", type=int, default=8)
    ap.add_argument("--vit-batch", type=int, default=64,
                    help="336x336 images per vision-tower call of the ViT throughput line (16 = the workload of rounds 1-4 is reported "
                         "next to it; 68 images, where every GEMM fills its last round of 256 workgroups, measured +0.4 %: the tail of a "
                         "launch is not a whole round)")
    ap.add_argument("--workload", default="qwen2vl-2b", choices=["qwen2vl-2b", "nanollava", "qwen2vl-7b-b32", "qwen2vl-2b-w4", "phi35v-w4-b16", "idefics2-b8"])
    ap.add_argument("--kv-bits", type=int, default=0, help="phi35v-w4-b16: 8 = uniform 8-bit KV cache (kv_bits of the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-hf", action="store_true", help="skip the HuggingFace torch-CPU second opinion of cpu_baseline")
    ap.add_argument("--no-extras", action="store_true", help="skip kernel rooflines / ViT throughput (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of the other BASELINE configs in the default line")
    ap.add_argument("--stage", default="", choices=["", "headline", "extras"],
                    help="internal: one stage of the default single-GPU line in a process of its own (see orchestrate)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank path: launcher, rendezvous (gloo), weight-arena broadcast, request deal, "
                         "timing protocol, result gather, JSON line - no kernels (tests/test_host_cpu.py runs it with --gpus 2)")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` outside torchrun: launch the N ranks ourselves (one process per GPU over RCCL, the form
        # the driver uses: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        print(f"[bench] --gpus {args.gpus} outside torchrun: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
        sys.exit(subprocess.call(cmd, env=env))
    ws_env = int(os.environ.get("WORLD_SIZE", "1"))
    if ws_env != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws_env}: the rank count of the launcher and --gpus must agree "
                 "(run `python bench.py --gpus N`, or torchrun --nproc-per-node N bench.py --gpus N)")

    from mlx_vlm_amd import parallel, synthetic

    if args.dry_run:
        rank, ws, local = parallel.init(backend="gloo")
        out = dry_run(args, rank, ws)
        if rank == 0:
            print(json.dumps(out), flush=True)
        parallel.barrier()
        parallel.shutdown()
        return
    ws = ws_env
    if args.workload != "qwen2vl-2b":
        rank, ws, local = parallel.init()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if rank == 0 and ws > 1:
            print(f"[bench] {ws} ranks, backend {_dist_info(ws)['backend']} (RCCL over xGMI), one process per GPU", file=sys.stderr, flush=True)
        out = {"nanollava": workload_nanollava, "qwen2vl-7b-b32": workload_7b_b32, "qwen2vl-2b-w4": workload_2b_w4,
               "phi35v-w4-b16": workload_phi35v_w4_b16, "idefics2-b8": workload_idefics2_b8}[args.workload](
            args, rank, ws, dev)
        kind = {"qwen2vl-7b-b32": "qwen2vl-7b", "qwen2vl-2b-w4": "qwen2vl-2b-w4", "phi35v-w4-b16": "phi35v-w4",
                "idefics2-b8": "idefics2-8b"}.get(args.workload)
        if kind:
            out = _with_cpu_baseline(out, kind, args, rank, ws)
        if rank == 0:
            print(json.dumps(out), flush=True)
        parallel.barrier()
        parallel.shutdown()
        return
    args.max_tokens = args.max_tokens or 256
    if args.stage == "headline" or ws > 1:
        # one rank of the N-rank job (torchrun / the self-launch above), or the headline child of the single-GPU orchestrator
        rank, ws, local = parallel.init()
        out = stage_headline(args, rank, ws, local)
        if rank == 0:
            if not args.stage:
                out.pop("_traffic_gate_up", None)          # (the orchestrator's hand-over field)
            print(json.dumps(out), flush=True)
        parallel.barrier()
        parallel.shutdown()
        return
    if args.stage == "extras":
        print(json.dumps(stage_extras(args)), flush=True)
        return
    orchestrate(args)

def _child(args, stage, timeout_s):
    """Run one stage of the default line as `bench.py --stage <stage>` -> (last JSON line or None, returncode, stderr tail)."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--stage", stage, "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--max-tokens", str(args.max_tokens), "--lookahead", str(args.lookahead), "--vit-batch", str(args.vit_batch)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        rc, so, se = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        def _txt(x):
            return x.decode(errors="replace") if isinstance(x, bytes) else (x or "")
        rc, so, se = -9, _txt(e.stdout), _txt(e.stderr) + f"\n[bench] stage {stage}: timeout after {timeout_s} s"
    lines = [ln for ln in so.strip().splitlines() if ln.startswith("{")]
    doc = None
    for ln in reversed(lines):
        try:
            doc = json.loads(ln)
            break
        except ValueError:          # a line cut short by an abort
            continue
    return doc, rc, se.strip()[-600:]

def orchestrate(args):
    """`python bench.py` on one GPU: this process never touches the GPU.  The headline (timed region) runs in a child whose JSON
    is echoed to stderr the moment it exists; the extras, the CPU baseline and the other configs follow, each in a process of
    its own, and ONE enriched line goes to stdout at the end.  A headline child killed by a signal is re-run (up to 3 attempts,
    every failed attempt is reported in `headline_attempts`)."""
    attempts = []
    out = None
    for _ in range(3):
        with ClockSampler() as clk:
            out, rc, err = _child(args, "headline", 900)
        if out is not None and rc == 0:
            break
        attempts.append({"rc": rc, "stderr_tail": err[-300:]})
        print(f"[bench] headline attempt {len(attempts)} failed (rc {rc}): {err[-300:]}", file=sys.stderr, flush=True)
        out = None
    if out is None:
        sys.exit(f"bench.py: the headline stage failed {len(attempts)} times: {attempts}")
    if attempts:
        out["headline_attempts"] = {"failed": attempts, "succeeded_on": len(attempts) + 1}
    # in `config` (which the driver's record keeps): a re-run headline must be impossible to miss
    out.setdefault("config", {})["headline_retries"] = len(attempts)
    out["config"]["vit_batch"] = args.vit_batch
    traffic_gu = out.pop("_traffic_gate_up", None)
    out["gpu_clocks"] = clk.summary()
    print("[bench] headline: " + json.dumps(out), file=sys.stderr, flush=True)

    extras = None
    if not args.no_extras:
        extras, rc, err = _child(args, "extras", 900)
        if rc != 0:
            extras = dict(extras or {})
            extras["error"] = f"extras stage rc {rc}: {err[-300:]}"
            print(f"[bench] extras stage failed (rc {rc}): {err[-300:]}", file=sys.stderr, flush=True)
    if extras:
        if extras.get("kernels"):
            k = extras["kernels"]["gemv_gate_up_swiglu"]
            out["roofline_kernel"] = {"bound": "hbm", "kernel": GATE_UP_KERNEL + " (RMSNorm + gate/up GEMV + SwiGLU, 28 launches/token)",
                                      "achieved": k["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["GBps"] / HBM_PEAK_GBS,
                                      "traffic": traffic_gu, "bytes_per_launch": k["bytes_per_launch"], "us_per_launch": k["us_per_launch"]}
            out["kernel_rooflines"] = extras["kernels"]
        if extras.get("vit336"):
            ips336, dt336 = extras["vit336"]
            ips448, dt448 = extras["vit448"]
            out["vision_images_per_s"] = ips336
            out["roofline_vit"] = {"bound": "mfma", "achieved": ips336 * VIT_TFLOP_336, "peak": MFMA_BF16_PEAK_TF,
                                   "unit": "TFLOP/s", "frac": ips336 * VIT_TFLOP_336 / MFMA_BF16_PEAK_TF, "traffic": None,
                                   "workload": f"{args.vit_batch} x 336x336 images per call ({args.vit_batch * 576} patches)",
                                   "ms_per_call": dt336 * 1e3}
            out["vision_single_448_images_per_s"] = ips448
            out["vision_single_448_tflops"] = ips448 * VIT_TFLOP_448
            out["vision_batch_sweep_336"] = extras.get("vit336_sweep")
            # the images/s half of BASELINE's metric INSIDE `roofline` (the object the driver's record keeps whole), like for like:
            # the quoted batch, the 16-image workload of rounds 1-4, and the single 448 x 448 image of configs[1] - all against
            # the same 2.5 PF (reference definition of the timed call: mlx_vlm/models/qwen2_vl/vision.py:257-290)
            def _vit(ips, ms, batch, tflop):
                return {"images_per_s": ips, "frac": ips * tflop / MFMA_BF16_PEAK_TF, "batch": batch, "ms_per_call": ms,
                        "achieved_tflops": ips * tflop, "peak_tflops": MFMA_BF16_PEAK_TF}
            out["roofline"]["vit"] = _vit(ips336, dt336 * 1e3, args.vit_batch, VIT_TFLOP_336)
            s16 = (extras.get("vit336_sweep") or {}).get("16")
            if args.vit_batch == 16:
                out["roofline"]["vit_16"] = dict(out["roofline"]["vit"])
            elif s16:
                out["roofline"]["vit_16"] = _vit(s16["images_per_s"], s16.get("ms_per_call"), 16, VIT_TFLOP_336)
            out["roofline"]["vit_single_448"] = _vit(ips448, dt448 * 1e3, 1, VIT_TFLOP_448)
            # what the matrix cores sustain on random operands with nothing else running (profiles/r06_gemm_power_limit.txt:
            # the chip is power-limited there - 1.87 PF at 1.85 GHz for v_mfma_f32_16x16x32_bf16, 2.37 PF on zeros)
            out["roofline"]["vit"]["frac_of_random_operand_mfma_ceiling"] = ips336 * VIT_TFLOP_336 / MFMA_RANDOM_OPERAND_TF
        for src, dst in (("batch8", "batch8_decode"), ("batch16", "batch16_decode"), ("wide64", "wide64_decode"),
                         ("continuous", "continuous_batching"), ("sampled", "sampled_decode")):
            out[dst] = extras.get(src)
        if extras.get("error"):
            out["extras_error"] = extras["error"]
    if not args.no_cpu_baseline:
        try:
            from mlx_vlm_amd.utils import cpu_quota
            out["cpu_baseline"] = cpu_baseline(min(cpu_quota(), 32), with_hf=not args.no_cpu_hf)          # the cores the container may use
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_extras and not args.no_configs:
        try:
            out["configs"] = other_configs(args, 0, 1, None, T_PROCESS_START)
        except Exception as e:
            out["configs"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)

if __name__ == "__main__":
    main()