"""Build libvlm_hip.so (the gfx950 operator library) in-tree with hipcc.

    python mlx-vlm_amd/build.py [--force]

Sources: mlx-vlm_amd/csrc/*.hip  ->  mlx-vlm_amd/lib/libvlm_hip.so
hipcc cross-compiles for gfx950 without a GPU; the .so travels to the GPU box
with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libvlm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators in architectural VGPRs instead of AGPRs.  hipcc's default put the flash
# attention accumulators (O^T, S^T) in AGPRs and then copied all 48 O^T registers to VGPRs at the top of EVERY key tile
# (v_accvgpr_read, for the rare rescale branch) - a full drain of the previous tile's P.V MFMAs; with the VGPR form the
# D = 80 kernel needs 218 registers instead of 172 + 80, D = 128 fits two waves per SIMD (212 vs 292), and the 128-tile
# GEMM loses its 5472 accumulator moves.  Results are bit-identical (profiles/r02_attn_prefill_probe.txt).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-pass-failed", "-mllvm", "-amdgpu-mfma-vgpr-form"]
# measurement builds only (e.g. VLM_BUILD_DEFINES="VLM_GEMM_ABLATION GEMM_STAMPS"): never set for the shipped library
FLAGS += ["-D" + d for d in os.environ.get("VLM_BUILD_DEFINES", "").split()]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "vlm_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    inc = [os.path.join(CSRC, m) for m in re.findall(r'#include "([^"/]+\.hip)"', open(src).read())]      # a TU that includes another
    newest = max([os.path.getmtime(src), _deps_mtime()] + [os.path.getmtime(i) for i in inc if os.path.exists(i)])
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
