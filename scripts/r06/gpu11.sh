#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_11; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "tiled" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 600 python3 scripts/r06/mlp_pair_bench.py 16 8 5 > $O/bench.out 2>&1; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -12 $O/pytest.out; cat $O/bench.out
