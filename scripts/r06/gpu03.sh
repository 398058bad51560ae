#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_03; mkdir -p $O
for abl in 0 1 2 3 4; do
  VLM_GEMM_F_ABL=$abl timeout 120 python3 scripts/r06/gemm_clock.py 8192 8192 8192 16 2>/dev/null | tail -1 >> $O/clock.out
done
VLM_GEMM_F_ABL=0 timeout 120 python3 scripts/r06/gemm_clock.py 8192 8192 8192 16 zero 2>/dev/null | tail -1 >> $O/clock.out
VLM_GEMM_F_ABL=0 timeout 120 python3 scripts/r06/gemm_clock.py 36864 5120 1280 16 2>/dev/null | tail -1 >> $O/clock.out
VLM_GEMM_F_ABL=0 timeout 120 python3 scripts/r06/gemm_clock.py 4096 4096 4096 16 2>/dev/null | tail -1 >> $O/clock.out
cat $O/clock.out
