#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_epi; mkdir -p $O
OLD=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip_old.so
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
for rep in 1 2; do
  for epi in gelu bias; do
    GEMM_SHAPES=vit GEMM_EPI=$epi timeout 300 python3 scripts/gemm_bench.py 3 > $O/new_${epi}_$rep.out 2>&1
    VLM_HIP_LIB=$OLD GEMM_SHAPES=vit GEMM_EPI=$epi timeout 300 python3 scripts/gemm_bench.py 3 > $O/old_${epi}_$rep.out 2>&1
  done
  timeout 300 python3 scripts/r05_vit_sweep.py 16 64 > $O/sweep_new_$rep.out 2>&1
  VLM_HIP_LIB=$OLD timeout 300 python3 scripts/r05_vit_sweep.py 16 64 > $O/sweep_old_$rep.out 2>&1
done
cat $O/rc.txt; tail -2 $O/pytest.out
for f in new_gelu_1 old_gelu_1 new_gelu_2 old_gelu_2 new_bias_1 old_bias_1 new_bias_2 old_bias_2; do echo "== $f"; grep mode3 $O/$f.out; done
for f in sweep_new_1 sweep_old_1 sweep_new_2 sweep_old_2; do echo "== $f"; tail -1 $O/$f.out | cut -c1-330; done
