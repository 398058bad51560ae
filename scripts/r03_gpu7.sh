#!/bin/bash
# Round-3 GPU session 7: q8 attention on the fp16 MFMA (tests) + Phi-3.5 b16 with / without the 8-bit KV cache; W4 GEMM tests.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s7
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_kv_quant_gpu.py -q --tb=short -k "q8 or kv_quantize or gemm_w4 or quantized or kv_bits" 2>&1 | grep -v "^$" | tail -30 ) > $O/t.log 2>&1; tail -12 $O/t.log
timeout 400 python bench.py --workload phi35v-w4-b16 --kv-bits 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_phi_kv8.json 2> $O/bench_phi_kv8.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_s7/bench_phi_kv8.json").read().strip().splitlines()[-1]); print("kv8", d["value"], d["roofline"]["frac"])
PY
