#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python3 -m pytest tests/test_engine_gpu.py tests/test_parity_decode_gpu.py tests/test_full_depth_gpu.py -q -m gpu -x 2>&1 | tail -3
for t in 0 1 0 1; do VLM_WIDE_TAILS=$t python3 bench.py --stage headline --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tails=$t ttft_ms', d['prefill_ms_to_first_token'], 'tok/s', d['value'])"; done
