#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_gemm; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm256" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
GEMM_SHAPES=vit GEMM_EPI=gelu timeout 300 python3 scripts/gemm_bench.py 3 10 14 > $O/bench_gelu.out 2>&1; echo "bench_gelu rc=$?" >> $O/rc.txt
GEMM_SHAPES=vit GEMM_EPI=bias timeout 300 python3 scripts/gemm_bench.py 3 10 14 > $O/bench_bias.out 2>&1; echo "bench_bias rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -5 $O/pytest.out; cat $O/bench_gelu.out $O/bench_bias.out
