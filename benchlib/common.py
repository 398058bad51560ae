"""Shared pieces of bench.py: roofline denominators, the synthetic request of BASELINE configs[1], one timed pass of the hot path,
synthetic model loading, the PMC-traffic record and the GPU clock sampler."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

__all__ = ['ROOT', 'T_PROCESS_START', 'HBM_PEAK_GBS', 'MFMA_BF16_PEAK_TF', 'MFMA_RANDOM_OPERAND_TF', 'VIT_TFLOP_448', 'VIT_TFLOP_336', 'build_request', 'run_step', 'time_events', 'DECODE_KERNELS', 'HEAD_KERNEL', 'GATE_UP_KERNEL', 'DECODE_CSRC', 'decode_csrc_sha16', 'pmc_traffic', '_load_synthetic', '_dist_info', 'ClockSampler']


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


T_PROCESS_START = time.perf_counter()


HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak


MFMA_RANDOM_OPERAND_TF = 1870.0   # measured (profiles/r06_gemm_power_limit.txt): bare v_mfma_f32_16x16x32_bf16 on N(0,1) operands, all CUs,


                                  # power-limited at 1.85 GHz (2.37 PF on zero operands) - an information line, never the roofline peak
VIT_TFLOP_448 = 1.481        # SURVEY.md §8d per 448^2 image


VIT_TFLOP_336 = 0.791


def build_request(cfg, image_hw, n_text, seed):
    from mlx_vlm_amd.models.qwen2_vl.processing_qwen2_vl import Qwen2VLImageProcessor

    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (3, image_hw, image_hw), dtype=np.uint8)
    out = Qwen2VLImageProcessor()([img])
    pix, thw = out["pixel_values"], out["image_grid_thw"]
    n_img = int(thw.prod()) // 4
    text = np.random.default_rng(1000 + seed).integers(0, 151643, n_text)
    ids = np.concatenate([[cfg.vision_start_token_id], np.full(n_img, cfg.image_token_id), [cfg.vision_start_token_id + 1], text])
    return ids.astype(np.int64)[None], torch.from_numpy(pix), thw


def run_step(model, req, max_tokens, lookahead):
    """-> (seconds to first token, seconds for the remaining tokens, tokens)"""
    from mlx_vlm_amd.generate import generate_step

    ids, pix, thw = req
    t0 = time.perf_counter()
    gen = generate_step(ids, model, pix, None, max_tokens=max_tokens, temperature=0.0, image_grid_thw=thw,
                        return_logprobs=False, lookahead=lookahead)
    toks, t_first = [], None
    for tok, _ in gen:
        if t_first is None:
            t_first = time.perf_counter()
        toks.append(tok)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return t_first - t0, t1 - t_first, toks


def time_events(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


DECODE_KERNELS = ("gemv_rowwave_kernel", "gemv_splitk_kernel", "attn_decode_mfma_kernel", "attn_decode_pagesplit_kernel",
                  "attn_decode_combine_kernel", "lse_partial_kernel", "logprob_argmax_kernel", "argmax_final_kernel",
                  "embed_gather_kernel", "decode_advance_kernel", "sample_filter_kernel", "logprob_argmax_tail_kernel")


HEAD_KERNEL = "gemv_rowwave_kernel<4, 3, 1, 1, 0>"         # RMSNorm + lm_head GEMV: exactly one launch per decoded token


GATE_UP_KERNEL = "gemv_rowwave_kernel<4, 3, 1, 1, 16>"     # name as rocprofv3 prints it (R=4 rows/wave, RMSNorm prologue, SwiGLU)


DECODE_CSRC = ("gemv_bf16.hip", "attn_decode.hip", "attn_pagesplit.hpp", "sample.hip", "embed.hip", "engine.hip", "common.hpp",
               "internal.h")


def decode_csrc_sha16():
    """hash of the sources of every kernel in the decode step + the engine that sequences them: what a PMC pass was taken on"""
    import hashlib

    h = hashlib.sha256()
    base = os.path.join(ROOT, "mlx-vlm_amd", "csrc")
    for f in DECODE_CSRC:
        h.update(open(os.path.join(base, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per launch from the committed --pmc passes (scripts/r04_final.sh -> profiles/r04_pmc_traffic.json):
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of MI355X_MICROARCH.md.  bench.py cannot collect hardware
    counters itself (they need rocprofv3 around the process).  The file records the hash of the decode step's kernel sources
    it was taken on (`_meta.decode_csrc_sha16`, scripts/pmc_summary.py); a file without it or with another hash is STALE and
    refused: -> (None, None, reason)."""
    here = os.path.join(ROOT, "profiles")
    cands = sorted((f for f in os.listdir(here) if f.endswith("_pmc_traffic.json")), reverse=True) if os.path.isdir(here) else []
    if not cands:
        return None, None, "no profiles/*_pmc_traffic.json"
    path = os.path.join(here, cands[0])
    d = json.load(open(path))
    sha = d.get("_meta", {}).get("decode_csrc_sha16")
    if sha != decode_csrc_sha16():
        return None, None, (f"{cands[0]} is stale: taken on decode sources {sha}, this tree is {decode_csrc_sha16()} "
                            "(re-run scripts/r04_final.sh)")
    gu = d.get(GATE_UP_KERNEL, {}).get("hbm_bytes_per_launch")
    steps = d.get(HEAD_KERNEL, {}).get("launches", 0) or d.get("decode_advance_kernel", {}).get("launches", 0)
    per_tok = None
    if steps:
        per_tok = sum(v.get("hbm_bytes_per_launch", 0.0) * v["launches"] for k, v in d.items()
                      if k.startswith(DECODE_KERNELS)) / steps
    return gu, per_tok, cands[0]


def _load_synthetic(cfg_dict, model_pkg, rank, dev, w4=False, **engine_kw):
    """rank 0 materialises the synthetic replica, the others receive it over RCCL/xGMI (parallel.broadcast_weights).
    w4: the language model as an MLX affine 4-bit checkpoint (random nibbles / scales / biases of that layout)."""
    from mlx_vlm_amd import parallel, synthetic
    from mlx_vlm_amd.utils import fit_host_threads, freeze_heap

    cfg = model_pkg.ModelConfig.from_dict(dict(cfg_dict))
    t0 = time.perf_counter()
    W = synthetic.random_weights(cfg, seed=0, device=dev, fill=(rank == 0))
    if w4 and getattr(cfg, "model_type", "") == "phi3_v":
        synthetic.quantize_random_(W, prefix="", skip=("model.vision_embed_tokens.",))
    elif w4:
        synthetic.quantize_random_(W)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    parallel.broadcast_weights(W, src=0)
    torch.cuda.synchronize()
    bcast_s = parallel.max_over_ranks(time.perf_counter() - t1, dev)
    nbytes = sum(v.numel() * v.element_size() for v in W.values())
    model = model_pkg.Model(cfg, device=dev, **engine_kw)
    model.load_weights(W)
    del W
    torch.cuda.synchronize()
    freeze_heap()          # what load() does: no 100 ms cyclic-GC passes over the import heap inside timed loops
    fit_host_threads()     # what load() does: torch's CPU pool capped at the container's CPU quota
    return cfg, model, {"load_s": time.perf_counter() - t0, "weight_bytes": nbytes, "broadcast_s": bcast_s}


def _dist_info(ws, load=None):
    import torch.distributed as dist

    out = {"backend": None, "ranks": 1}
    if ws > 1 and dist.is_initialized():
        out = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}
        if load and load.get("broadcast_s"):
            out["weight_broadcast_GBps"] = load["weight_bytes"] / load["broadcast_s"] / 1e9
    return out


class ClockSampler:
    """GPU clocks while a config's child process runs (the Phi-3.5 4-bit line read 3450 vs 4200 tok/s in two
    contexts - clock / thermal state or the process?).  A thread samples the current shader and memory clock levels from the
    amdgpu sysfs tables (`pp_dpm_sclk` / `pp_dpm_mclk`: the line with the `*`), falling back to one `rocm-smi -c --json` call
    before and after when the tables are not there.  The parent never opens the device for this."""

    def __init__(self, period_s=0.25):
        import glob
        self.period = period_s
        self.paths = {}
        # which card is ours: a node's sysfs lists all eight, rocm-smi only the one this container was given
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        if cards:
            mine = [k for k in (self._smi_once() or {}).get("_cards", []) if os.path.isdir(f"/sys/class/drm/{k}/device")]
            for card in ([f"/sys/class/drm/{mine[0]}/device"] if mine else cards):
                if os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                    self.paths = {"sclk": os.path.join(card, "pp_dpm_sclk"), "mclk": os.path.join(card, "pp_dpm_mclk")}
                    break
        self.samples = {"sclk": [], "mclk": []}
        self.smi = []
        self._stop = None
        self._thread = None

    @staticmethod
    def _current_mhz(path):
        try:
            for ln in open(path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        except Exception:
            return None
        return None

    @staticmethod
    def _smi_once():
        import subprocess
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-c", "--json"], capture_output=True, text=True, timeout=15)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            out = {k.strip(" :"): v for k, v in card.items() if "sclk" in k or "mclk" in k}
            out["_cards"] = sorted(d)
            return out
        except Exception as e:
            return {"error": f"{type(e).__name__}: {e}"}

    def __enter__(self):
        import threading
        if not self.paths:
            self.smi.append(self._smi_once())
            return self
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                for k, pth in self.paths.items():
                    v = self._current_mhz(pth)
                    if v is not None:
                        self.samples[k].append(v)
                self._stop.wait(self.period)

        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2)
        else:
            self.smi.append(self._smi_once())
        return False

    def summary(self):
        if self._thread is None:
            return {"source": "rocm-smi -c --json, before / after", "before": self.smi[0] if self.smi else None,
                    "after": self.smi[1] if len(self.smi) > 1 else None}
        out = {"source": "%s, sampled every %.2f s" % (self.paths.get("sclk", ""), self.period)}
        for k, v in self.samples.items():
            if v:
                sv = sorted(v)
                out[k + "_mhz"] = {"min": sv[0], "median": sv[len(sv) // 2], "max": sv[-1], "n": len(sv)}
        return out
