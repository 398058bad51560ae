#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_22; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_full_depth_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "wide" > $O/pytest.out 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.out
for t in 1 0; do
  VLM_WIDE_TAILS=$t timeout 600 python3 bench.py --workload qwen2vl-7b-b32 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tails=$t', d['value'], json.dumps(d['roofline'])[:300])"
done
