#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_tail; mkdir -p $O
timeout 900 python3 -m pytest tests/test_engine_gpu.py tests/test_sampler_gpu.py tests/test_ops_gpu.py tests/test_parity_decode_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest.out 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.out
timeout 200 python3 bench.py --stage extras --gpus 1 --steps 2 --warmup 1 2> $O/extras.err | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('sampled', {k: round(v['decode_us_per_token'], 1) for k, v in d.get('sampled').items()})
"
