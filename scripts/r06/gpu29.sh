#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_29; mkdir -p $O
(cd /tmp && rm -rf /tmp/p_v && timeout -k 15 300 rocprofv3 --kernel-trace --stats -d /tmp/p_v -o p -- python3 $R/scripts/vit_prof.py 1 448 20 > $O/prof.log 2>&1); echo "prof rc=$?"
python3 $R/scripts/prof_summary.py $(find /tmp/p_v -name "*.db" | head -1) $O/stats_vit448.txt > /dev/null 2>&1
grep 'images' $O/prof.log; head -16 $O/stats_vit448.txt
