#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_05; mkdir -p $O
for sh in "8192 8192 8192 7" "8192 8192 8192 7 zero" "36864 5120 1280 7" "9216 5120 1280 7" "36864 1280 5120 6" "4096 4096 4096 7"; do
  timeout 120 python3 scripts/r06/gemm_clock3.py $sh 2>/dev/null | tail -1 >> $O/clock3.out
done
cat $O/clock3.out
