#!/bin/bash
# kernel-trace stats of the Phi-3.5 W4 b16 and 7B b32 workloads
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_21; mkdir -p $O
for w in phi35v-w4-b16 qwen2vl-7b-b32; do
  (cd /tmp && rm -rf /tmp/p_$w && timeout -k 15 500 rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o p -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 > $O/prof_$w.log 2>&1); echo "prof $w rc=$?"
  python3 $R/scripts/prof_summary.py $(find /tmp/p_$w -name "*.db" | head -1) $O/stats_$w.txt > /dev/null 2>&1
  grep '^{' $O/prof_$w.log | cut -c1-600
  head -16 $O/stats_$w.txt
done
