#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call2; mkdir -p $O
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench.out 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python3 -m pytest tests/test_bench_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "bench or greedy_advance or driver or headline" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 600 python3 scripts/r05_vit_sweep.py > $O/vit_sweep.out 2> $O/vit_sweep.err; echo "sweep rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -3 $O/pytest.out; tail -1 $O/vit_sweep.out
