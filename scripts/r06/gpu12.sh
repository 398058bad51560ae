#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_12; mkdir -p $O
for k in 1 0 1 0; do echo "VLM_GEMV_MFMA_ROWS=$k" >> $O/batch_ab.out; VLM_GEMV_MFMA_ROWS=$k timeout 600 python3 scripts/r06/batch_ab.py 16 8 2>/dev/null | grep "^rows" >> $O/batch_ab.out; done
timeout 2400 python3 -m pytest tests/test_engine_gpu.py tests/test_full_depth_gpu.py tests/test_ops_gpu.py -x -q -m gpu > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cat $O/rc.txt; cat $O/batch_ab.out; tail -5 $O/pytest.out
