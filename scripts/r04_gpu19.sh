#!/bin/bash
# Round-4 GPU session 19: long-K form on the o_proj shapes too (K = 3584 / 4096, 224 / 256 tiles)?
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu19
mkdir -p $O
cd $R
for lk in 4096 2048; do
  echo "== VLM_GEMV_MFMA_LONGK=$lk"
  VLM_GEMV_MFMA_LONGK=$lk timeout 300 python scripts/mfma_shapes.py 7b mistral --rows 16,8 2>&1 | grep -E "==|o_proj|down|layer"
done
