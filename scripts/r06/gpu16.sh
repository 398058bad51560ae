#!/bin/bash
# kernel-trace stats of the headline child on the current tree
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_16; mkdir -p $O
(cd /tmp && rm -rf /tmp/p1 && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python3 $R/bench.py --stage headline --steps 3 --warmup 1 > $O/prof.log 2>&1); echo "prof rc=$?"
python3 $R/scripts/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/r06_headline_kernel_stats.txt > /dev/null 2>&1
head -14 $O/r06_headline_kernel_stats.txt
