#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_modes; mkdir -p $O
for rep in 1 2; do
  GEMM_SHAPES=vit GEMM_EPI=bias timeout 300 python3 scripts/gemm_bench.py 3 4 7 > $O/bias_$rep.out 2>&1
done
timeout 300 python3 scripts/gemm_bench.py 3 4 > $O/std.out 2>&1
cat $O/bias_1.out $O/bias_2.out $O/std.out | grep mode
