#!/bin/bash
# A/B two environment settings: bench (decode tok/s) + rocprof kernel durations of the decode kernels.
# usage: scripts/ab_env.sh <tag> "<VAR=VAL ...>" [<tag> "<VAR=VAL ...>" ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
while [ $# -ge 2 ]; do
  tag=$1; envs=$2; shift 2
  echo "== $tag ($envs)"
  env $envs timeout 200 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tok/s',d['value'],'ms/step',d['ms_per_step'])"
  (cd /tmp && rm -rf /tmp/prof_$tag && env $envs timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/prof_$tag.log 2>&1)
  db=$(find /tmp/prof_$tag -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $db $R/gpurun_out/ab_$tag.txt > /dev/null 2>&1
  grep -E "gemv|attn_decode|lse|argmax|embed|advance|copyBuffer" $R/gpurun_out/ab_$tag.txt | cut -c1-130
done
