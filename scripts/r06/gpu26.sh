#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_26; mkdir -p $O
(cd /tmp && rm -rf /tmp/p_w && timeout -k 15 400 rocprofv3 --kernel-trace --stats -d /tmp/p_w -o p -- python3 $R/scripts/r06/wide_prof.py 64 > $O/prof.log 2>&1); echo "prof rc=$?"
python3 $R/scripts/prof_summary.py $(find /tmp/p_w -name "*.db" | head -1) $O/stats_wide64.txt > /dev/null 2>&1
grep '^{' $O/prof.log; head -22 $O/stats_wide64.txt
