#!/bin/bash
# Round-4 GPU session 16: long-K form - activations through LDS (XLDS) vs fragment-shaped loads; tests
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu16
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemv_mfma_rows_plain" 2>&1 | tail -3
for x in 1 0; do
  echo "== VLM_GEMV_MFMA_LONGK_XLDS=$x"
  VLM_GEMV_MFMA_LONGK_XLDS=$x timeout 300 python scripts/mfma_shapes.py 7b mistral --rows 16,8 > $O/shapes_xlds$x.txt 2>&1
  grep -v "amdgpu.ids\|knobs" $O/shapes_xlds$x.txt | grep -E "==|down|layer" | head -40
done
echo "== XLDS=1, all 2B tiles too (MIN_TILES=64)"
VLM_GEMV_MFMA_LONGK_MIN_TILES=64 timeout 300 python scripts/mfma_shapes.py 2b --rows 16,8 2>&1 | grep -E "==|down|layer"
