#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_topp4; mkdir -p $O
timeout 600 python3 -m pytest tests/test_sampler_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "sampl or split or gumbel or top_p or categorical" > $O/pytest.out 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 200 python3 bench.py --stage extras --gpus 1 --steps 2 --warmup 1 2> $O/extras.err | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('sampled', d.get('sampled'))
"
timeout 120 python3 bench.py --stage headline --gpus 1 --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('greedy', d['value'], d['decode_us_per_token'] if 'decode_us_per_token' in d else '')
"
cat $O/rc.txt; tail -4 $O/pytest.out
