#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_topp3; mkdir -p $O
timeout 600 python3 -m pytest tests/test_sampler_gpu.py -x -q -m gpu > $O/pytest.out 2>&1; echo "pytest rc=$?" > $O/rc.txt
(cd /tmp && rm -rf /tmp/pt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python3 $GRAFT_REPO_ROOT/scripts/r05_topp_prof.py > $O/prof.log 2>&1)
python3 scripts/prof_summary.py $(find /tmp/pt -name "*.db" | head -1) $O/kernels.txt > /dev/null 2>&1
cat $O/rc.txt; tail -3 $O/pytest.out; grep -E "topp|gumbel|argmax|lse|filter|kernel  " $O/kernels.txt | cut -c1-140; tail -5 $O/prof.log
