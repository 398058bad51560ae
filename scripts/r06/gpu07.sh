#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_07; mkdir -p $O
timeout 900 python3 -m pytest tests/test_op_noise_gpu.py -q -m gpu -s > $O/noise.out 2>&1; echo "noise rc=$?" >> $O/rc.txt
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attn_decode" > $O/ops.out 2>&1; echo "ops rc=$?" >> $O/rc.txt
cat $O/rc.txt; grep -E "HIP vs exact|passed|failed|Error" $O/noise.out; tail -4 $O/ops.out
