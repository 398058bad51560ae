#!/bin/bash
# Round-3 third evidence run: wide decode steps (17..64 rows on the prefill GEMMs).  GPU suite, the decode forward of 9 / 16 /
# 32 / 57 rows against the oracle, the Qwen2-VL-7B batch-32 line at 32 rows per GPU (with its CPU baseline) and its kernel stats.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final3
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_full_depth_gpu.py --tb=line 2>&1 | grep -v "^$" | tail -30 ) > $O/t_all.log 2>&1; tail -6 $O/t_all.log
( time timeout 400 python -m pytest tests/test_full_depth_gpu.py -q -s --tb=line -k decode_forward 2>&1 | grep -E "decode forward|passed|failed|FAILED|Error" ) > $O/t_rows.log 2>&1; cat $O/t_rows.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py --workload qwen2vl-7b-b32 --steps 2 --warmup 1 > $O/r03_bench_qwen2vl-7b-b32_c.json 2> $O/bench_7b.err; tail -c 700 $O/r03_bench_qwen2vl-7b-b32_c.json; echo
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_7b -o s -- python $R/bench.py --workload qwen2vl-7b-b32 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_7b.log 2>&1; echo "prof7b rc=$?"
cd $R
python scripts/prof_summary.py $(find $O/prof_7b -name "*.db" | head -1) $O/r03_7b_b32_kernel_stats_wide.txt | head -16
rm -rf $O/prof_7b
