#!/bin/bash
# Session 13: hand-counted waits (assembly loads) + 3 sets in flight + K split for every form - tests, per-projection times,
# the 16 / 8-row 2B steps.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s13
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "gemv_mfma or gemv_w4_mfma" 2>&1 | tail -25 ) > $O/t_mfma.log 2>&1; tail -6 $O/t_mfma.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python scripts/mfma_shapes.py 2b 7b mistral phi-w4 --rows 16 > $O/shapes_$tag.txt 2>&1
  grep -E "^==|layer|qkv norm\+rope|qkv norm\+bias|o_proj|gate|down" $O/shapes_$tag.txt | awk '/^==/ {printf "%s:", $0; next} /layer/ {printf " layer %s us %s TB/s\n", $5, $7; next} {printf " %s %s |", $1, $(NF-3)}' | sed "s/^/[$tag] /"
}
run asm VLM_GEMV_MFMA_DEBUG=1
run asm_seg8 VLM_GEMV_MFMA_SEG_CHUNKS=8
run asm_wg3 VLM_GEMV_MFMA_WGS_PER_CU=3
run asm_wg1 VLM_GEMV_MFMA_WGS_PER_CU=1
run asm_nosplit VLM_GEMV_MFMA_NORM_SPLIT=0
timeout 300 python scripts/batch_prof.py 16 64 2>&1 | tail -1
timeout 300 python scripts/batch_prof.py 8 64 2>&1 | tail -1
grep "gemv_mfma\]" $O/shapes_asm.txt | sort -u | head -30
