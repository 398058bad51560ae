#!/bin/bash
# Session 18: wide decode steps (32 / 64 rows on the prefill GEMMs) - tests and the 7B batch-32 job at 16 vs 32 rows.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s18
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_engine_gpu.py -q --tb=short -k "batch_generator or generate_batch" 2>&1 | tail -25 ) > $O/t_wide.log 2>&1; tail -12 $O/t_wide.log
for rows in 16 32; do
  VLM_BENCH_7B_ROWS=$rows timeout 400 python bench.py --workload qwen2vl-7b-b32 --steps 2 --warmup 1 --no-cpu-baseline > $O/7b_rows$rows.json 2> $O/7b_rows$rows.err
  python - <<P
import json
try:
    d=json.loads(open("$O/7b_rows$rows.json").read().strip().splitlines()[-1])
    print("7B rows=$rows e2e", round(d["value"],1), "decode", round(d["decode_tokens_per_s"],1), "frac", round(d["roofline"]["frac"],4), "ms/job", round(d["ms_per_step"],1), "steps", d["roofline"]["decode_steps"])
except Exception as e:
    print("7B rows=$rows failed", e); print(open("$O/7b_rows$rows.err").read()[-1500:])
P
done
