#!/bin/bash
# A/B two builds of the operator library: bench (decode tok/s) + rocprof kernel durations of the decode kernels.
# usage: scripts/ab_prof.sh <tag> <lib.so> [<tag> <lib.so> ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
while [ $# -ge 2 ]; do
  tag=$1; export VLM_HIP_LIB=$2; shift 2
  echo "== $tag ($VLM_HIP_LIB)"
  timeout 200 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tok/s',d['value'],'ms/step',d['ms_per_step'])"
  (cd /tmp && rm -rf /tmp/prof_$tag && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/prof_$tag.log 2>&1)
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $db $R/gpurun_out/ab_$tag.txt > /dev/null 2>&1
  grep -E "gemv|attn_decode|lse|argmax|embed|advance" $R/gpurun_out/ab_$tag.txt | cut -c1-130
done
