#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_19
mkdir -p $O
export TMPDIR=/tmp
for w in top_p chain; do
  (cd /tmp && rm -rf /tmp/p_$w && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o s -- python3 $R/scripts/r06/sampled_one.py $w > $O/log_$w.txt 2>&1)
  python3 $R/scripts/prof_summary.py $(find /tmp/p_$w -name "*.db" | head -1) $O/stats_$w.txt > /dev/null 2>&1
  grep -i "topp\|sample\|gumbel\|lse_partial\|logprob\|advance\|Name" $O/stats_$w.txt | head -20
  grep '^{' $O/log_$w.txt
done
