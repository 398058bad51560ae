#!/bin/bash
# Session 15: all register sets requested before the activation prologue - tests, per-projection times, batch lines.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s15
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "gemv_mfma or gemv_w4_mfma" 2>&1 | tail -15 ) > $O/t_gemv.log 2>&1; tail -3 $O/t_gemv.log
timeout 300 python scripts/mfma_shapes.py 2b 7b mistral phi-w4 --rows 16 > $O/shapes.txt 2>&1
grep -E "^==|layer|qkv norm\+rope|qkv norm\+bias|o_proj|gate|down" $O/shapes.txt | awk '/^==/ {printf "%s:", $0; next} /layer/ {printf " layer %s us %s TB/s\n", $5, $7; next} {printf " %s %s |", $1, $(NF-3)}'
timeout 300 python scripts/batch_prof.py 16 64 2>&1 | tail -1
timeout 300 python scripts/batch_prof.py 8 64 2>&1 | tail -1
