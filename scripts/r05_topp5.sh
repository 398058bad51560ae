#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_topp5; mkdir -p $O
(cd /tmp && rm -rf /tmp/pt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 $GRAFT_REPO_ROOT/scripts/r05_topp_prof2.py > $O/prof.log 2>&1)
python3 scripts/prof_summary.py $(find /tmp/pt -name "*.db" | head -1) $O/kernels.txt > /dev/null 2>&1
grep -E "topp|gumbel|argmax|lse|logprob|filter|kernel  " $O/kernels.txt | cut -c1-140
