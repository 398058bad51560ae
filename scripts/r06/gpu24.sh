#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for t in 1 2 3 1 2 3; do
  VLM_GEMM_SKINNY_NO256=$t timeout 600 python3 bench.py --workload qwen2vl-7b-b32 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no256=$t', d['value'], d['roofline']['frac'])"
done
