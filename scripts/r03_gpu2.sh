#!/bin/bash
# Round-3 GPU session 2: o_proj-prologue merge, DPP reductions, marginal costs, kernel profile, full-depth parity, whole suite.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s2
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "paged_split or attn_decode or gemv or library" 2>&1 | tail -5 > $O/t_ops.log; cat $O/t_ops.log
P=scripts/bin/decode_probe
V="--variant 0,256,0x7f,1,96"
timeout 300 $P --steps 300 --ctx 450 --no-hot $V,0,0,0,0,1 $V,0,0,16,1,0 $V,0,0,16,1,1 $V,0,0,8,1,1 $V,0,0,16,0,1 $V,0,0,16,3,1 > $O/probe_ctx450.txt 2>&1; cat $O/probe_ctx450.txt
timeout 300 $P --steps 200 --ctx 450 --no-hot $V,0,0,16,1,1 $V,0x01,0,16,1,1 $V,0x02,0,16,1,1 $V,0x04,0,16,1,1 $V,0x08,0,16,1,1 $V,0x10,0,16,1,1 $V,0x20,0,16,1,1 $V,0x40,0,16,1,1 $V,0x1f,0,16,1,1 > $O/probe_marginal.txt 2>&1; cat $O/probe_marginal.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof.log 2>&1; echo "prof rc=$?"
cd $R
python scripts/prof_summary.py $(find $O/prof -name "*.db" | head -1) $O/kernel_stats.txt | head -16
( time timeout 1500 python -m pytest tests/test_full_depth_gpu.py -q -s 2>&1 | grep -v "^$" | tail -25 ) > $O/t_full.log 2>&1; cat $O/t_full.log
( time timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_full_depth_gpu.py 2>&1 | tail -12 ) > $O/t_all.log 2>&1; cat $O/t_all.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
