#!/bin/bash
# Round-4 GPU session 15: long-K form of the decode GEMM (gemv_mfma_longk.hip): operator tests, per-projection A/B
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu15
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemv_mfma_rows_plain" 2>&1 | tail -4
for lk in 0 4096 2048; do
  echo "== VLM_GEMV_MFMA_LONGK=$lk"
  VLM_GEMV_MFMA_LONGK=$lk timeout 300 python scripts/mfma_shapes.py 2b 7b mistral --rows 16,8 > $O/shapes_lk$lk.txt 2>&1
  grep -v "amdgpu.ids\|knobs" $O/shapes_lk$lk.txt | grep -E "==|o_proj|down|layer" | head -40
done
