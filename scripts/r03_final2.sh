#!/bin/bash
# Round-3 second evidence run, after the batched-decode GEMM work (csrc/gemv_mfma.hip) and the BatchGenerator time accounting fix:
# the GPU suite, the 9 / 16-row decode forward vs the oracle, smoke, kernel stats of the 16-row 2B step and of the 7B batch-32
# job, and the bench lines whose numbers those changes move (the one-row decode step and its PMC passes are untouched:
# profiles/r03_pmc_traffic.json still carries the hash of the sources it was taken on).  -> gpurun_out/r03_final2/
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final2
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_full_depth_gpu.py --tb=line 2>&1 | grep -v "^$" | tail -30 ) > $O/t_all.log 2>&1; tail -6 $O/t_all.log
( time timeout 400 python -m pytest tests/test_full_depth_gpu.py -q -s --tb=line -k decode_forward 2>&1 | grep -E "decode forward|passed|failed|FAILED|Error" ) > $O/t_rows.log 2>&1; cat $O/t_rows.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof_b16 -o s -- python $R/scripts/batch_prof.py 16 64 > $O/prof_b16.log 2>&1; echo "prof16 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_7b -o s -- python $R/bench.py --workload qwen2vl-7b-b32 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_7b.log 2>&1; echo "prof7b rc=$?"
cd $R
python scripts/prof_summary.py $(find $O/prof_b16 -name "*.db" | head -1) $O/r03_batch16_kernel_stats_b.txt | head -8
python scripts/prof_summary.py $(find $O/prof_7b -name "*.db" | head -1) $O/r03_7b_b32_kernel_stats.txt | head -10
rm -rf $O/prof_b16 $O/prof_7b
timeout 500 python bench.py --steps 5 --warmup 2 > $O/r03_bench_line_b.json 2> $O/bench.err; tail -c 900 $O/r03_bench_line_b.json; echo
for w in qwen2vl-7b-b32 idefics2-b8; do
  timeout 400 python bench.py --workload $w --steps 2 --warmup 1 > $O/r03_bench_${w}_b.json 2> $O/bench_$w.err; echo "$w rc=$?"; tail -c 500 $O/r03_bench_${w}_b.json; echo
done
for w in phi35v-w4-b16 qwen2vl-2b-w4; do
  timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_bench_${w}_b.json 2> $O/bench_$w.err; echo "$w rc=$?"; tail -c 400 $O/r03_bench_${w}_b.json; echo
done
timeout 300 python bench.py --workload phi35v-w4-b16 --kv-bits 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_bench_phi35v-w4-b16-kv8_b.json 2> $O/bench_phi_kv8.err; tail -c 300 $O/r03_bench_phi35v-w4-b16-kv8_b.json; echo
