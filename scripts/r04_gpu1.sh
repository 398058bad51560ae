#!/bin/bash
# Round-4 GPU session 1: the whole GPU suite on the cleaned library (ABI v5) with the new parity tests (wide rows at 7B widths,
# wide x 8-bit K/V, batch kv policy, Python callables, thinking budgets, full-depth greedy identity), smoke, the default bench
# line, and the "profile first" runs of VERDICT item 5: kernel stats of the 2B 4-bit one-row step and of the Phi-3.5 kv8 step.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu1
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_full_depth_gpu.py --tb=short -x 2>&1 | grep -v "^$" | tail -40 ) > $O/t_all.log 2>&1; tail -8 $O/t_all.log
( time timeout 900 python -m pytest tests/test_full_depth_gpu.py -q -s --tb=short 2>&1 | grep -E "rel-rms|identical|passed|failed|FAILED|Error|assert" | tail -60 ) > $O/t_full.log 2>&1; tail -25 $O/t_full.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; echo
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_w4 -o s -- python $R/bench.py --workload qwen2vl-2b-w4 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_w4.log 2>&1; echo "prof w4 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_kv8 -o s -- python $R/bench.py --workload phi35v-w4-b16 --kv-bits 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_kv8.log 2>&1; echo "prof kv8 rc=$?"
cd $R
python scripts/prof_summary.py $(find $O/prof_w4 -name "*.db" | head -1) $O/r04_w4_kernel_stats.txt | head -14
python scripts/prof_summary.py $(find $O/prof_kv8 -name "*.db" | head -1) $O/r04_phi35v_kv8_kernel_stats.txt | head -14
rm -rf $O/prof_w4 $O/prof_kv8
