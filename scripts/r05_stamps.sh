#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_stamps; mkdir -p $O
export VLM_HIP_LIB=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip_stamps.so
for args in "36864 5120 1280 gelu" "36864 5120 1280 bias" "9216 5120 1280 gelu" "36864 1280 5120 bias" "8192 8192 8192 none"; do
  timeout 120 python3 scripts/r05_gemm_stamps.py $args >> $O/stamps.txt 2>> $O/err.txt
done
cat $O/stamps.txt; tail -5 $O/err.txt
