#!/bin/bash
# 4-bit decode: rocprof kernel summary of the bench command
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_w4 -o w4 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload qwen2vl-2b-w4 --steps 2 --warmup 1 > /tmp/w4_prof.log 2>&1
tail -5 /tmp/w4_prof.log | cut -c1-300
find /tmp/prof_w4 -type f | head
f=$(find /tmp/prof_w4 -name "*kernel_stats.csv" | head -1)
head -25 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r02_w4_kernel_stats.csv
cut -c1-220 $GRAFT_REPO_ROOT/gpurun_out/r02_w4_kernel_stats.csv
