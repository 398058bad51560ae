#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_gelu; mkdir -p $O
OLD=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip_old.so
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
for rep in 1 2 3; do
  GEMM_SHAPES=vit GEMM_EPI=gelu timeout 300 python3 scripts/gemm_bench.py 3 > $O/new_gelu_$rep.out 2>&1
  VLM_HIP_LIB=$OLD GEMM_SHAPES=vit GEMM_EPI=gelu timeout 300 python3 scripts/gemm_bench.py 3 > $O/old_gelu_$rep.out 2>&1
  timeout 300 python3 scripts/r05_vit_sweep.py 16 64 > $O/sweep_new_$rep.out 2>&1
  VLM_HIP_LIB=$OLD timeout 300 python3 scripts/r05_vit_sweep.py 16 64 > $O/sweep_old_$rep.out 2>&1
done
VLM_HIP_LIB=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip_stamps.so timeout 120 python3 scripts/r05_gemm_stamps.py 36864 5120 1280 gelu > $O/stamps.txt 2>> $O/err.txt
cat $O/rc.txt; tail -2 $O/pytest.out
for rep in 1 2 3; do echo "== rep $rep new | old"; paste <(grep mode3 $O/new_gelu_$rep.out) <(grep mode3 $O/old_gelu_$rep.out | awk '{print $5,$6,$7,$8}'); done
for rep in 1 2 3; do echo "new: $(tail -1 $O/sweep_new_$rep.out | cut -c1-200)"; echo "old: $(tail -1 $O/sweep_old_$rep.out | cut -c1-200)"; done
cat $O/stamps.txt
