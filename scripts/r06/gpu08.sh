#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_08; mkdir -p $O
timeout 900 python3 -m pytest tests/test_op_noise_gpu.py -q -m gpu -s > $O/noise.out 2>&1; echo "noise rc=$?" >> $O/rc.txt
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_kv_quant_gpu.py -x -q -m gpu > $O/ops.out 2>&1; echo "ops rc=$?" >> $O/rc.txt
timeout 600 python3 bench.py --stage headline --steps 3 --warmup 1 > $O/headline.json 2> $O/headline.err; echo "headline rc=$?" >> $O/rc.txt
cat $O/rc.txt; grep -E "HIP vs exact|passed|failed|Error" $O/noise.out | cut -c1-150; tail -4 $O/ops.out; python3 -c "
import json; d=json.load(open('$O/headline.json')); print(d['value'], d['decode_us_per_token'], d['roofline']['frac'])"
