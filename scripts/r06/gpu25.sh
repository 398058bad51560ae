#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_25; mkdir -p $O
timeout 3000 python3 -m pytest tests -q -m gpu -x > $O/pytest.out 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest.out
for w in qwen2vl-7b-b32 phi35v-w4-b16 idefics2-b8 nanollava; do
for t in 0 1; do
  VLM_GEMM_SKINNY64=$t VLM_WIDE_TAILS=$t timeout 600 python3 bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w new=$t', d['value'], d['roofline']['frac'], d.get('prompt_tps'))"
done; done 2>&1 | tee $O/ab.txt
