#!/bin/bash
# Round-4 GPU session 17: half-page units of the 8-bit-KV decode attention: operator + engine tests (forced on), Phi-3.5 kv8 A/B
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu17
mkdir -p $O
cd $R
echo "== q8 tests with half pages forced"
VLM_ATTN_Q8_HALF=1 timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_kv_quant_gpu.py tests/test_full_depth_gpu.py -q -x -k "q8 or quant or kv_bits or kv8" 2>&1 | tail -4
echo "== default policy"
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_kv_quant_gpu.py -q -x -k "q8 or quant or kv_bits" 2>&1 | tail -3
for h in 1 0; do
  echo "== VLM_ATTN_Q8_HALF=$h"
  VLM_ATTN_Q8_HALF=$h timeout 300 python bench.py --workload phi35v-w4-b16 --kv-bits 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tok/s',d['value'],'frac',d['roofline']['frac'],'e2e',d.get('e2e_tokens_per_s'))"
done
(cd /tmp && rm -rf /tmp/prof_kv8 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kv8 -o p -- python $R/bench.py --workload phi35v-w4-b16 --kv-bits 8 --no-cpu-baseline > /tmp/prof_kv8.log 2>&1)
db=$(find /tmp/prof_kv8 -name "*.db" | head -1)
python scripts/prof_summary.py $db $O/phi35v_kv8_half.txt > /dev/null 2>&1
head -6 $O/phi35v_kv8_half.txt | cut -c1-150; grep kv_quantize $O/phi35v_kv8_half.txt | cut -c1-150
