#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_23; mkdir -p $O
w=qwen2vl-7b-b32
(cd /tmp && rm -rf /tmp/p_$w && timeout -k 15 500 rocprofv3 --kernel-trace --stats -d /tmp/p_$w -o p -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 2 --warmup 1 > $O/prof_$w.log 2>&1); echo "prof $w rc=$?"
python3 $R/scripts/prof_summary.py $(find /tmp/p_$w -name "*.db" | head -1) $O/stats_$w.txt > /dev/null 2>&1
head -24 $O/stats_$w.txt
