#!/bin/bash
# Round-3 end-of-round evidence run on the GPU box (one gpurun call): the whole GPU suite, the bench lines of every workload
# (roofline + cpu_baseline each), rocprofv3 kernel summaries (default bench, 16-row step, end-of-round ViT) and the two --pmc
# passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) behind roofline.traffic.  Everything lands in
# gpurun_out/r03_final/; the summaries worth keeping are copied into profiles/ afterwards.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_final
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_full_depth_gpu.py --tb=line 2>&1 | grep -v "^$" | tail -40 ) > $O/t_all.log 2>&1; tail -8 $O/t_all.log
( time timeout 1500 python -m pytest tests/test_full_depth_gpu.py -q -s --tb=line 2>&1 | grep -E "full-depth|decode forward|passed|failed|FAILED|Error" ) > $O/t_full.log 2>&1; cat $O/t_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
# ---- HBM traffic (two PMC passes) and kernel stats of the default bench
SHORT="python $R/bench.py --steps 1 --warmup 0 --max-tokens 12 --no-cpu-baseline --no-extras"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $SHORT > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- $SHORT > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python scripts/pmc_summary.py $O/r03_pmc_traffic.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) | head -10
cp $O/r03_pmc_traffic.json $R/profiles/r03_pmc_traffic.json     # (so that the bench lines below carry roofline.traffic)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1; echo "prof rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b16 -o s -- python $R/scripts/batch_prof.py 16 64 > $O/prof_b16.log 2>&1; echo "prof16 rc=$?"
timeout 250 rocprofv3 --kernel-trace --stats -d $O/prof_vit -o v -- python $R/scripts/vit_prof.py 16 > $O/prof_vit.log 2>&1; echo "vitprof rc=$?"
cd $R
python scripts/prof_summary.py $(find $O/prof_bench -name "*.db" | head -1) $O/r03_bench_kernel_stats.txt | head -12
python scripts/prof_summary.py $(find $O/prof_b16 -name "*.db" | head -1) $O/r03_batch16_kernel_stats.txt | head -14
python scripts/prof_summary.py $(find $O/prof_vit -name "*.db" | head -1) $O/r03_vit16_kernel_stats.txt | head -10
rm -rf $O/pmc_fetch $O/pmc_write $O/prof_bench $O/prof_b16 $O/prof_vit
# ---- the bench lines
timeout 600 python bench.py --steps 5 --warmup 2 > $O/r03_bench_line.json 2> $O/bench.err; tail -c 1200 $O/r03_bench_line.json; echo
for w in nanollava qwen2vl-2b-w4 qwen2vl-7b-b32 idefics2-b8 phi35v-w4-b16; do
  timeout 600 python bench.py --workload $w --steps 2 --warmup 1 > $O/r03_bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; tail -c 600 $O/r03_bench_$w.json; echo
done
timeout 600 python bench.py --workload phi35v-w4-b16 --kv-bits 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_bench_phi35v-w4-b16-kv8.json 2> $O/bench_phi_kv8.err; tail -c 600 $O/r03_bench_phi35v-w4-b16-kv8.json; echo
