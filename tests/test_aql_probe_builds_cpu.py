"""scripts/aql_probe (the clean-room launch chain behind profiles/r05_aql_fence_probe.txt) still builds: the kernels cross-compile
for gfx950 into a code object without hidden kernel arguments (the probe fills none), the host program compiles and links against
the HSA runtime.  No GPU is touched."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "scripts", "aql_probe")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("g++") is None, reason="needs hipcc and g++")
def test_probe_kernels_and_host_program_build(tmp_path):
    hsaco = str(tmp_path / "probe.hsaco")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "--offload-device-only", "--no-gpu-bundle-output", "-O3", "-o", hsaco,
                        os.path.join(SRC, "probe_kernels.hip")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", hsaco], capture_output=True, text=True, timeout=60).stdout
    for k in ("k_empty", "k_chain_plain", "k_chain_agent", "k_gemv_agent", "k_link_barrier", "k_link_flag"):
        assert f".name:           {k}" in notes, k
    assert "hidden_" not in notes                      # no implicit arguments: the host side writes explicit ones only
    exe = str(tmp_path / "aql_probe")
    r = subprocess.run(["g++", "-O2", "-I/opt/rocm/include", "-o", exe, os.path.join(SRC, "aql_probe.cpp"), "-L/opt/rocm/lib",
                        "-lhsa-runtime64", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.getsize(exe) > 0
