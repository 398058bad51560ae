#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_gaps; mkdir -p $O
(cd /tmp && rm -rf /tmp/prof_gaps && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_gaps -o p -- python3 $GRAFT_REPO_ROOT/bench.py --stage headline --gpus 1 --steps 2 --warmup 1 > $O/run.log 2>&1)
db=$(find /tmp/prof_gaps -name "*.db" | head -1)
python3 scripts/r05_decode_gaps.py $db $O/decode_gaps.txt > /dev/null 2> $O/gaps.err
python3 scripts/prof_summary.py $db $O/kernel_stats.txt > /dev/null 2>&1
head -30 $O/decode_gaps.txt; tail -3 $O/gaps.err
