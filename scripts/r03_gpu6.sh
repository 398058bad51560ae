#!/bin/bash
# Round-3 GPU session 6: W4 fused GEMM + native tower loop validation, family suites, workload benches (first look).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s6
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "gemm_w4 or dequant" 2>&1 | grep -v "^$" | tail -30 ) > $O/t_w4.log 2>&1; tail -12 $O/t_w4.log
( time timeout 900 python -m pytest tests/test_vlm_family_idefics2_gpu.py tests/test_vlm_family_llava_bunny_gpu.py tests/test_vlm_family_phi3v_gpu.py tests/test_parity_decode_gpu.py -q --tb=line 2>&1 | grep -v "^$" | tail -30 ) > $O/t_fam.log 2>&1; tail -14 $O/t_fam.log
for w in idefics2-b8 phi35v-w4-b16 nanollava; do
  timeout 400 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; tail -c 1500 $O/bench_$w.json; echo
done
timeout 400 python bench.py --workload phi35v-w4-b16 --kv-bits 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_phi_kv8.json 2> $O/bench_phi_kv8.err; tail -c 1500 $O/bench_phi_kv8.json; echo
