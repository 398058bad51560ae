#!/bin/bash
# Round-2 evidence run on the GPU box: HBM traffic (two --pmc passes) and kernel stats of the default bench workload.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp
SHORT="python $R/bench.py --steps 1 --warmup 0 --max-tokens 12 --no-cpu-baseline --no-extras"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $SHORT > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- $SHORT > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python scripts/pmc_summary.py $O/r02_pmc_traffic.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) | head -14
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_final -o r02 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_final.log 2>&1; echo "prof rc=$?"
cd $R
python scripts/prof_summary.py $(find $O/prof_final -name "*.db" | head -1) $O/r02_bench_kernel_stats_final.txt | head -16
