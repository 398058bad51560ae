#!/bin/bash
# End-of-round evidence run on the GPU box: full bench line, kernel-trace profile of the bench, ViT profile, and the
# separate --pmc pass (FETCH_SIZE / WRITE_SIZE) for the HBM-traffic figure.  Everything lands in gpurun_out/.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-r01}
timeout 420 python $R/bench.py --steps 3 --warmup 1 > $O/bench_$TAG.log 2> $O/bench_$TAG.err; echo "bench rc=$?"; tail -c 2500 $O/bench_$TAG.log
(cd /tmp && rm -rf /tmp/p1 && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_$TAG.log 2>&1); echo "prof rc=$?"
python $R/scripts/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/${TAG}_bench_kernel_stats.txt > /dev/null 2>&1
(cd /tmp && rm -rf /tmp/p2 && timeout 250 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o v -- python $R/scripts/vit_prof.py 16 > $O/vitprof_$TAG.log 2>&1); echo "vitprof rc=$?"
python $R/scripts/prof_summary.py $(find /tmp/p2 -name "*.db" | head -1) $O/${TAG}_vit16_kernel_stats.txt > /dev/null 2>&1
# FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 (and the refused run hangs until its timeout): one pass each
bash $R/scripts/r06/evidence_pmc.sh
