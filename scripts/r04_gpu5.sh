#!/bin/bash
# Round-4 GPU session 5: gemv_mfma2 with 4 register sets (4-bit) / 2 (short bf16 K): tests + per-projection A/B of the unit policy
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu5
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "mfma" 2>&1 | tail -4
for mu in 384 200; do
  VLM_GEMV_MFMA2_MIN_UNITS=$mu timeout 300 python scripts/mfma_shapes.py 2b 7b mistral phi-w4 --rows 16 > $O/shapes_mu$mu.txt 2>&1
  echo "== MIN_UNITS=$mu"; grep -v "gemv_mfma\|amdgpu.ids\|knobs" $O/shapes_mu$mu.txt | tail -26
done
timeout 300 python scripts/mfma_shapes.py 2b phi-w4 --rows 8 > $O/shapes_rows8.txt 2>&1; grep -v "gemv_mfma\|amdgpu.ids\|knobs" $O/shapes_rows8.txt | tail -12
