"""bench.py's single-GPU orchestrator on the host (no GPU): a headline child that dies is re-run and reported, a dying extras child
costs only the extras it had not finished, and exactly ONE JSON line reaches stdout (BENCH_r04 was a GPU memory
fault that erased the whole line)."""
import argparse
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(**over):
    d = dict(gpus=1, steps=2, warmup=1, max_tokens=256, lookahead=8, vit_batch=64, no_cpu_baseline=True, no_cpu_hf=True, no_extras=False,
             no_configs=True, stage="")
    d.update(over)
    return argparse.Namespace(**d)


HEAD = {"metric": "decode tokens/sec + vision-prefill images/sec, Qwen2-VL-2B", "value": 1100.0, "unit": "tokens/s", "n_gpus": 1,
        "roofline": {"bound": "hbm", "frac": 0.43}, "_traffic_gate_up": 55e6, "decode_nan_rows": 0}
EXTRAS = {"kernels": {"gemv_gate_up_swiglu": {"GBps": 5200.0, "bytes_per_launch": 55050240, "us_per_launch": 10.5}},
          "vit336": (1240.0, 0.0516), "vit448": (210.0, 0.00476), "vit336_sweep": {"16": {"images_per_s": 1150.0}},
          "batch8": {"generation_tps": 5800.0}}


def test_headline_child_is_retried_and_one_line_is_printed(monkeypatch, capsys):
    import bench

    calls = []

    def fake_child(args, stage, timeout_s):
        calls.append(stage)
        if stage == "headline":
            if calls.count("headline") == 1:
                return None, -6, "Memory access fault by GPU node-2"
            return dict(HEAD), 0, ""
        return dict(EXTRAS), 0, ""

    monkeypatch.setattr(bench, "_child", fake_child)
    bench.orchestrate(_args())
    out, err = capsys.readouterr()
    lines = [ln for ln in out.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and calls == ["headline", "headline", "extras"]
    d = json.loads(lines[0])
    assert d["value"] == 1100.0 and "_traffic_gate_up" not in d
    assert d["headline_attempts"]["succeeded_on"] == 2 and d["headline_attempts"]["failed"][0]["rc"] == -6
    assert "Memory access fault" in d["headline_attempts"]["failed"][0]["stderr_tail"]
    assert d["roofline_kernel"]["traffic"] == 55e6 and d["roofline_vit"]["workload"].startswith("64 x 336x336")
    assert abs(d["roofline_vit"]["frac"] - 1240.0 * bench.VIT_TFLOP_336 / 2500.0) < 1e-12
    # the images/s half of the metric inside `roofline` (what the driver's record keeps whole), and a retry impossible to miss
    assert d["roofline"]["vit"]["batch"] == 64 and abs(d["roofline"]["vit"]["frac"] - d["roofline_vit"]["frac"]) < 1e-12
    assert d["roofline"]["vit_16"]["batch"] == 16 and abs(d["roofline"]["vit_16"]["images_per_s"] - 1150.0) < 1e-9
    assert d["roofline"]["vit_single_448"]["batch"] == 1 and d["roofline"]["vit_single_448"]["images_per_s"] == 210.0
    assert d["config"]["headline_retries"] == 1 and d["config"]["vit_batch"] == 64
    assert d["batch8_decode"] == {"generation_tps": 5800.0} and d["wide64_decode"] is None
    assert "[bench] headline: {" in err and "headline attempt 1 failed" in err


def test_extras_fault_keeps_the_headline_and_the_finished_extras(monkeypatch, capsys):
    import bench

    def fake_child(args, stage, timeout_s):
        if stage == "headline":
            return dict(HEAD), 0, ""
        return {"kernels": EXTRAS["kernels"]}, -6, "Memory access fault"         # died after the first extra

    monkeypatch.setattr(bench, "_child", fake_child)
    bench.orchestrate(_args())
    d = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert d["value"] == 1100.0 and "headline_attempts" not in d
    assert d["kernel_rooflines"] == EXTRAS["kernels"] and "roofline_vit" not in d and "extras stage rc -6" in d["extras_error"]


def test_three_dead_headline_children_end_the_run(monkeypatch):
    import bench

    monkeypatch.setattr(bench, "_child", lambda args, stage, timeout_s: (None, -11, "boom"))
    with pytest.raises(SystemExit) as e:
        bench.orchestrate(_args(no_extras=True))
    assert "failed 3 times" in str(e.value)


def test_clock_sampler_reads_the_starred_level_and_degrades_without_sysfs(tmp_path):
    """bench.ClockSampler: the current level of an amdgpu `pp_dpm_*` table is the line that ends with `*`; without the tables
    (this container) the summary says so instead of raising."""
    import time

    import bench

    sclk = tmp_path / "pp_dpm_sclk"
    sclk.write_text("S: 95Mhz\n0: 500Mhz\n1: 2100Mhz *\n2: 2400Mhz\n")
    mclk = tmp_path / "pp_dpm_mclk"
    mclk.write_text("0: 900Mhz\n1: 2000Mhz *\n")
    assert bench.ClockSampler._current_mhz(str(sclk)) == 2100
    c = bench.ClockSampler(period_s=0.01)
    c.paths = {"sclk": str(sclk), "mclk": str(mclk)}
    with c:
        time.sleep(0.05)
    s = c.summary()
    assert s["sclk_mhz"]["median"] == 2100 and s["mclk_mhz"]["max"] == 2000 and s["sclk_mhz"]["n"] >= 1
    d = bench.ClockSampler()
    d.paths = {}
    with d:
        pass
    assert "before" in d.summary() and "after" in d.summary()
