#!/bin/bash
# Round-4 GPU session 4: second form of the skinny-M decode GEMM (gemv_mfma2.hip) - operator tests, per-projection A/B
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu4
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "mfma" 2>&1 | tail -15
for v in 0 1; do
  VLM_GEMV_MFMA2=$v VLM_GEMV_MFMA_DEBUG=1 timeout 300 python scripts/mfma_shapes.py 2b 7b mistral phi-w4 --rows 16 > $O/shapes_v2_$v.txt 2>&1
  echo "== VLM_GEMV_MFMA2=$v"; grep -v "gemv_mfma\|amdgpu.ids" $O/shapes_v2_$v.txt | tail -30
done
