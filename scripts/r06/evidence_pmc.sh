#!/bin/bash
# HBM-traffic PMC passes over the headline stage alone (no child processes under the profiler).  FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950: two runs of the same command, both databases into scripts/pmc_summary.py.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ev; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rm -rf /tmp/p_$c && timeout -k 15 240 rocprofv3 --kernel-trace --pmc $c -d /tmp/p_$c -o c -- python3 $R/bench.py --stage headline --steps 1 --warmup 0 --max-tokens 24 > $O/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
done
python3 $R/scripts/pmc_summary.py $O/r06_pmc_traffic.json $(find /tmp/p_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/p_WRITE_SIZE -name "*.db" | head -1) 2>&1 | head -12
