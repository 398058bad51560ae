#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_33; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_ops_gpu.py tests/test_full_depth_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "gemm or wide" > $O/pytest.out 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.out
for t in 0 1 0 1; do
  VLM_GEMM_SKINNY32=$t timeout 600 python3 bench.py --workload qwen2vl-7b-b32 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skinny32=$t', d['value'], d['roofline']['frac'])"
done
