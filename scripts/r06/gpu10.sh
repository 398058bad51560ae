#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_10; mkdir -p $O
for rep in 1 2; do
for aux in 0 1 2 16 17; do
  L=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip.so; [ $aux != 0 ] && L=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip_aux$aux.so
  GEMM_SHAPES=vit GEMM_EPI=bias VLM_HIP_LIB=$L timeout 200 python3 scripts/gemm_bench.py 3 2>/dev/null | sed "s/^/aux=$aux /" >> $O/aux.out
  VLM_HIP_LIB=$L timeout 200 python3 scripts/r06/gemm_one.py 8192 8192 8192 3 2>/dev/null | sed "s/^/aux=$aux /" >> $O/aux.out
done; done
cat $O/aux.out
