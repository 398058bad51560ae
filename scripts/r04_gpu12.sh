#!/bin/bash
# Round-4 GPU session 12: coalesced vlm_kv_quantize_tokens - tests (bit-exact vs the oracle) and Phi-3.5 16 rows with the 8-bit KV cache
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu12
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kv_quant_gpu.py tests/test_ops_gpu.py -q -x -k "q8 or quant or kv8 or kv_q" 2>&1 | tail -4
for kv in 8 0; do
  echo "== --kv-bits $kv"
  timeout 300 python bench.py --workload phi35v-w4-b16 --kv-bits $kv --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tok/s',d['value'],'frac',d['roofline']['frac'],'e2e',d.get('e2e_tokens_per_s'))"
done
(cd /tmp && rm -rf /tmp/prof_kv8 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kv8 -o p -- python $R/bench.py --workload phi35v-w4-b16 --kv-bits 8 --no-cpu-baseline > /tmp/prof_kv8.log 2>&1)
db=$(find /tmp/prof_kv8 -name "*.db" | head -1)
python scripts/prof_summary.py $db $O/phi35v_kv8.txt > /dev/null 2>&1
head -12 $O/phi35v_kv8.txt | cut -c1-150
