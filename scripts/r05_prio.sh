#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_prio; mkdir -p $O
for rep in 1 2; do
  timeout 300 python3 scripts/r05_attn_bench.py > $O/base$rep.out 2>&1
  VLM_HIP_LIB=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip_prio.so timeout 300 python3 scripts/r05_attn_bench.py > $O/prio$rep.out 2>&1
done
VLM_HIP_LIB=$GRAFT_REPO_ROOT/mlx-vlm_amd/lib/libvlm_hip_prio.so timeout 300 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k attn_prefill > $O/pytest.out 2>&1
grep imgs $O/base*.out $O/prio*.out | cut -c1-160; tail -2 $O/pytest.out
