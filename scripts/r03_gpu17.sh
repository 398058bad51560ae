#!/bin/bash
# Session 17: would a decode step of more than 16 rows pay through the prefill GEMMs?  The four projections at 32 / 64 rows.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s17
mkdir -p $O
cd $R
timeout 300 python scripts/mfma_shapes.py 7b 2b mistral --rows "" --gemm-rows 32,64,17 > $O/gemm_rows.txt 2>&1; grep -v amdgpu.ids $O/gemm_rows.txt | tail -60
