#!/bin/bash
# Session 8: the row-trip activation prologue of the skinny-M MFMA decode GEMM (tests, batch lines before / after through the
# A/B knob), and where the Qwen2-VL-7B batch-32 job spends its end-to-end time (rocprofv3 kernel stats + host cProfile).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s8
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=line -k "gemv_mfma or gemv_w4_mfma or library" 2>&1 | tail -15 ) > $O/t_mfma.log 2>&1; tail -6 $O/t_mfma.log
for knob in 0 1; do
  VLM_GEMV_MFMA_TRIPS=$knob timeout 300 python scripts/batch_prof.py 16 64 > $O/b16_trips$knob.log 2>&1; tail -1 $O/b16_trips$knob.log
done
timeout 400 python bench.py --workload qwen2vl-7b-b32 --steps 2 --warmup 1 --no-cpu-baseline > $O/7b_trips1.json 2> $O/7b_trips1.err
python - <<P
import json
d=json.loads(open("$O/7b_trips1.json").read().strip().splitlines()[-1])
print("7B e2e", round(d["value"],1), "decode", round(d["decode_tokens_per_s"],1), "frac", round(d["roofline"]["frac"],4), "ms/job", round(d["ms_per_step"],1))
P
timeout 300 python scripts/profile_7b_host.py 32 64 > $O/host_7b.txt 2>&1; head -4 $O/host_7b.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_7b -o s -- python $R/bench.py --workload qwen2vl-7b-b32 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_7b.log 2>&1; echo "prof7b rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_phi -o s -- python $R/bench.py --workload phi35v-w4-b16 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_phi.log 2>&1; echo "profphi rc=$?"
cd $R
python scripts/prof_summary.py $(find $O/prof_7b -name "*.db" | head -1) $O/r03_7b_b32_kernel_stats.txt | head -24
python scripts/prof_summary.py $(find $O/prof_phi -name "*.db" | head -1) $O/r03_phi35v_kernel_stats.txt | head -16
rm -rf $O/prof_7b $O/prof_phi
