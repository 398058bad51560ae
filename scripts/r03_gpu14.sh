#!/bin/bash
# Session 14: the kept form of the batched decode GEMM (compiler-visible loads, row-trip prologue, deferred hand-off, measured
# K-segment policy): tests, per-projection times (-> profiles/r03_mfma_shapes.txt), batch lines.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s14
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "gemv" 2>&1 | tail -15 ) > $O/t_gemv.log 2>&1; tail -4 $O/t_gemv.log
VLM_GEMV_MFMA_DEBUG=1 timeout 300 python scripts/mfma_shapes.py 2b 7b mistral phi-w4 --rows 16,8 > $O/shapes.txt 2>&1
grep -E "^==|layer" $O/shapes.txt | awk '/^==/ {printf "%s:", $0; next} {printf " layer %s us %s TB/s\n", $5, $7}'
timeout 300 python scripts/batch_prof.py 16 64 2>&1 | tail -1
timeout 300 python scripts/batch_prof.py 8 64 2>&1 | tail -1
timeout 400 python bench.py --workload qwen2vl-7b-b32 --steps 2 --warmup 1 --no-cpu-baseline > $O/7b.json 2> $O/7b.err
python - <<P
import json
d=json.loads(open("$O/7b.json").read().strip().splitlines()[-1])
print("7B e2e", round(d["value"],1), "decode", round(d["decode_tokens_per_s"],1), "frac", round(d["roofline"]["frac"],4), "ms/job", round(d["ms_per_step"],1), "steps", d["roofline"]["decode_steps"], "decode_s", round(d["roofline"]["decode_time_s"],3))
P
