"""The driver's own command, executed on the GPU box as part of the suite (BENCH_r04 was a GPU memory
fault on a tree whose final .so had never run bench.py).  Every test here starts `bench.py` as the driver does - a process of
its own - and demands a parsed JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra, capture_output=True, text=True,
                       timeout=timeout, env=e, cwd=ROOT)
    assert r.returncode == 0, f"bench.py rc {r.returncode}\n{r.stderr[-2000:]}"
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"bench.py must print ONE JSON line, got {len(lines)}"
    return json.loads(lines[0]), r.stderr


def test_driver_command_default_line():
    """`python bench.py --gpus 1 --steps 2 --warmup 1` through the orchestrator: headline child + extras child; the line carries
    the roofline objects and no stage failed."""
    out, err = _run(["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-configs"])
    assert out["metric"].startswith("decode tokens/sec") and out["unit"] == "tokens/s" and out["n_gpus"] == 1
    assert out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 100.0
    assert "headline_attempts" not in out, out.get("headline_attempts")          # ANY retry of the headline child fails the suite
    assert out["config"]["headline_retries"] == 0 and out["config"]["vit_batch"] == 64
    assert "extras_error" not in out, out.get("extras_error")
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0.05 < rf["frac"] < 1.0 and rf["peak"] == 8000.0
    # the images/s half of the metric sits inside `roofline` (the object the driver's record keeps), like for like
    for key, batch in (("vit", 64), ("vit_16", 16), ("vit_single_448", 1)):
        v = rf[key]
        assert v["batch"] == batch and v["images_per_s"] > 10.0 and 0.02 < v["frac"] < 1.0 and v["peak_tflops"] == 2500.0, (key, v)
    assert abs(rf["vit"]["frac"] - out["roofline_vit"]["frac"]) < 1e-12
    assert 0.05 < out["roofline_vit"]["frac"] < 1.0 and out["roofline_kernel"]["us_per_launch"] > 1.0
    for k in ("batch8_decode", "batch16_decode", "wide64_decode", "continuous_batching", "sampled_decode"):
        assert out[k] and "error" not in out[k], (k, out[k])
    assert "[bench] headline: {" in err           # the headline is on stderr before any extra runs


@pytest.mark.parametrize("layout", ["identity", "paged"])
def test_headline_configuration_under_both_kv_layouts(layout):
    """The benchmark's engine (2B dims, kv_pool_tokens=32768, max_seqs=40: a 37 GB identity pool or the shared free list), a 448 x
    448 image + 128 text tokens, 256 greedy tokens, lookahead 8 - the configuration no other test builds."""
    out, _ = _run(["--stage", "headline", "--steps", "1", "--warmup", "1"], env={"VLM_KV_LAYOUT": layout})
    assert out["config"]["max_tokens"] == 256 and out["config"]["decode_lookahead"] == 8 and out["config"]["prompt_tokens"] == 386
    assert out["value"] > 100.0 and 0.05 < out["roofline"]["frac"] < 1.0


def test_headline_canary_with_one_hipmalloc_per_tensor():
    """The headline child once per suite with torch's caching allocator and the HSA fragment allocator off: every tensor is
    its own hipMalloc, so an out-of-bounds access of any kernel in the timed region lands on an unmapped page and kills the
    process instead of reading a neighbour (the canary for the unexplained memory fault of BENCH_r04)."""
    out, _ = _run(["--stage", "headline", "--steps", "1", "--warmup", "1"],
                  env={"PYTORCH_NO_CUDA_MEMORY_CACHING": "1", "HSA_DISABLE_FRAGMENT_ALLOCATOR": "1"})
    assert out["value"] > 100.0 and out["decode_nan_rows"] == 0 and out["config"]["prompt_tokens"] == 386
