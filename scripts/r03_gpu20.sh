#!/bin/bash
# Session 20: the two --pmc passes again (engine.hip changed - wide branch - so the committed traffic file is refused as stale;
# the one-row kernels are the same), then the 8-row batch line.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s20
mkdir -p $O
SHORT="python $R/bench.py --steps 1 --warmup 0 --max-tokens 12 --no-cpu-baseline --no-extras"
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $SHORT > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- $SHORT > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python scripts/pmc_summary.py $O/r03_pmc_traffic.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) | head -6
rm -rf $O/pmc_fetch $O/pmc_write
timeout 100 python scripts/batch_prof.py 8 64 2>&1 | tail -1
