#!/bin/bash
# Evidence pass A on the current tree: kernel-trace stats of the headline stage, the ViT at 16 / 64 images and the 16-row batched
# decode.  Summaries land in gpurun_out/r06_ev/ (copy the ones to keep into profiles/).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ev; mkdir -p $O
(cd /tmp && rm -rf /tmp/p1 && timeout -k 15 400 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python3 $R/bench.py --stage headline --steps 3 --warmup 1 > $O/prof.log 2>&1); echo "prof rc=$?"
python3 $R/scripts/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/r06_bench_kernel_stats.txt > /dev/null 2>&1
for nb in 16 64; do
  (cd /tmp && rm -rf /tmp/p2 && timeout -k 15 300 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o v -- python3 $R/scripts/vit_prof.py $nb > $O/vitprof_$nb.log 2>&1); echo "vitprof $nb rc=$?"
  python3 $R/scripts/prof_summary.py $(find /tmp/p2 -name "*.db" | head -1) $O/r06_vit${nb}_kernel_stats.txt > /dev/null 2>&1
done
(cd /tmp && rm -rf /tmp/p4 && timeout -k 15 300 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o b -- python3 $R/scripts/batch_prof.py 16 64 > $O/batchprof.log 2>&1); echo "batchprof rc=$?"
python3 $R/scripts/prof_summary.py $(find /tmp/p4 -name "*.db" | head -1) $O/r06_batch16_kernel_stats.txt > /dev/null 2>&1
grep "^B=" $O/batchprof.log
head -8 $O/r06_bench_kernel_stats.txt; head -8 $O/r06_vit16_kernel_stats.txt; head -8 $O/r06_vit64_kernel_stats.txt; head -12 $O/r06_batch16_kernel_stats.txt
