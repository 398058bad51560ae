#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_06; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_sampler_gpu.py tests/test_kv_quant_gpu.py -x -q -m gpu > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -15 $O/pytest.out
