#!/bin/bash
# Round-3 GPU session 4: suite after the test / engine fixes; TLB-touch experiment; early-sincos qkv; 3 big full-depth numbers.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s4
mkdir -p $O
cd $R
P=scripts/bin/decode_probe
V="--variant 0,256,0x7f,1,96"
timeout 300 $P --steps 300 --ctx 450 --no-hot $V,0,0,16,1,1,0 $V,0,0,16,1,1,1 $V,0,0,16,1,1,3 $V,0x02,0,16,1,1,0 $V,0x02,0,16,1,1,3 $V,0,0,16,1,1,0 > $O/probe_tlb.txt 2>&1; cat $O/probe_tlb.txt
timeout 300 $P --steps 200 --ctx 450 --no-hot $V,0x01,0,16,1,1,0 $V,0x04,0,16,1,1,0 $V,0x04,0,16,1,1,3 $V,0x10,0,16,1,1,0 $V,0x10,0,16,1,1,3 > $O/probe_marg.txt 2>&1; cat $O/probe_marg.txt
( time timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_full_depth_gpu.py --tb=line 2>&1 | grep -v "^$" | tail -60 ) > $O/t_all.log 2>&1; tail -40 $O/t_all.log
( time timeout 1500 python -m pytest tests/test_full_depth_gpu.py -q -s --tb=line -k "full_depth" 2>&1 | grep -E "row rel-rms|full-depth|passed|failed|FAILED|Error" | grep -v "B=" ) > $O/t_full.log 2>&1; cat $O/t_full.log
