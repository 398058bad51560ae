#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_01; mkdir -p $O
timeout 600 python3 scripts/r06/gemm_f_check.py > $O/check.out 2>&1; echo "check rc=$?" >> $O/rc.txt
timeout 300 python3 scripts/gemm_bench.py 3 14 > $O/bench.out 2>&1; echo "bench rc=$?" >> $O/rc.txt
GEMM_SHAPES=vit GEMM_EPI=gelu timeout 300 python3 scripts/gemm_bench.py 3 14 > $O/bench_vit_gelu.out 2>&1; echo "bench_vit rc=$?" >> $O/rc.txt
GEMM_SHAPES=vit GEMM_EPI=bias timeout 300 python3 scripts/gemm_bench.py 3 14 > $O/bench_vit_bias.out 2>&1
cat $O/rc.txt; tail -15 $O/check.out; cat $O/bench.out $O/bench_vit_gelu.out $O/bench_vit_bias.out
