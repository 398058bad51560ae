#!/bin/bash
# Round-4 GPU session 11: Phi-3.5 4-bit, 16 rows - end to end and per kernel with the second MFMA decode GEMM on / off
# (the default line of session 10 has this config 9 % BELOW the pre-mfma2 line although every projection measured faster alone)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu11
mkdir -p $O
cd $R
for v in 1 0; do
  export VLM_GEMV_MFMA2=$v
  echo "== VLM_GEMV_MFMA2=$v"
  timeout 300 python bench.py --workload phi35v-w4-b16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tok/s',d['value'],'frac',d['roofline']['frac'],'e2e',d.get('e2e_tokens_per_s'))"
  (cd /tmp && rm -rf /tmp/prof_$v && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p -- python $R/bench.py --workload phi35v-w4-b16 --no-cpu-baseline > /tmp/prof_$v.log 2>&1)
  db=$(find /tmp/prof_$v -name "*.db" | head -1)
  python scripts/prof_summary.py $db $O/phi35v_mfma2_$v.txt > /dev/null 2>&1
  head -14 $O/phi35v_mfma2_$v.txt | cut -c1-150
done
