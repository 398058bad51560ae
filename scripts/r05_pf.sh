#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_pf; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 600 python3 -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "tower or vit or vision or features" > $O/pytest_tower.out 2>&1; echo "pytest_tower rc=$?" >> $O/rc.txt
for rep in 1 2; do
  for epi in gelu bias; do GEMM_SHAPES=vit GEMM_EPI=$epi timeout 300 python3 scripts/gemm_bench.py 3 > $O/new_${epi}_$rep.out 2>&1; done
  timeout 300 python3 scripts/r05_vit_sweep.py 16 64 > $O/sweep_new_$rep.out 2>&1
done
export VLM_HIP_LIB=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip_stamps.so
for args in "36864 5120 1280 gelu" "36864 5120 1280 bias" "36864 3840 1280 bias"; do timeout 120 python3 scripts/r05_gemm_stamps.py $args >> $O/stamps.txt 2>> $O/err.txt; done
cat $O/rc.txt; tail -2 $O/pytest.out; tail -2 $O/pytest_tower.out
for f in new_gelu_1 new_gelu_2 new_bias_1 new_bias_2; do echo "== $f"; grep mode3 $O/$f.out; done
for f in sweep_new_1 sweep_new_2; do tail -1 $O/$f.out | cut -c1-330; done
cat $O/stamps.txt
