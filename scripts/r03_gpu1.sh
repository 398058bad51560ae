#!/bin/bash
# Round-3 GPU session 1: new page-split attention + full-depth parity tests + decode A/B probe + per-kernel profile.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s1
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "paged_split or attn_decode" 2>&1 | tail -5 > $O/t_ops.log; cat $O/t_ops.log
P=scripts/bin/decode_probe
V="--variant 0,256,0x7f,1,96,0,0"
timeout 300 $P --steps 300 --ctx 450 --no-hot $V,0,0 $V,16,0 $V,8,0 $V,32,0 $V,16,1 $V,16,2 $V,0,1 $V,0,0 > $O/probe_ctx450.txt 2>&1; cat $O/probe_ctx450.txt
timeout 200 $P --steps 200 --ctx 1500 --no-hot $V,0,0 $V,16,0 $V,32,0 > $O/probe_ctx1500.txt 2>&1; cat $O/probe_ctx1500.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_a -o a -- $R/$P --steps 60 --ctx 450 --no-hot $V,0,0 > $O/prof_a.log 2>&1; echo "prof a rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_b -o b -- $R/$P --steps 60 --ctx 450 --no-hot $V,16,1 > $O/prof_b.log 2>&1; echo "prof b rc=$?"
cd $R
for x in a b; do python scripts/prof_summary.py $(find $O/prof_$x -name "*.db" | head -1) $O/kernel_stats_$x.txt | head -14; done
( time timeout 1500 python -m pytest tests/test_full_depth_gpu.py -q -x -s 2>&1 | tail -15 ) > $O/t_full.log 2>&1; cat $O/t_full.log
( time timeout 900 python -m pytest tests -q -x -m gpu --deselect tests/test_full_depth_gpu.py 2>&1 | tail -8 ) > $O/t_all.log 2>&1; cat $O/t_all.log
