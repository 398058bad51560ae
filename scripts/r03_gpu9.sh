#!/bin/bash
# Session 9: per-projection times of the batched decode GEMVs at 2B / 7B / Mistral / Phi-3.5 (4-bit) dims under the policy knobs
# of csrc/gemv_mfma.hip (K segments of the norm-prologue forms, chunks per unit of the segment forms, workgroups per CU).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s9
mkdir -p $O
cd $R
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python scripts/mfma_shapes.py 2b 7b mistral phi-w4 --rows 16 > $O/shapes_$tag.txt 2>&1
  grep -E "^==|layer|qkv norm\+rope|qkv norm\+bias|o_proj|gate|down" $O/shapes_$tag.txt | awk '{printf "%s | ", $0} /layer/ {print ""}' | sed "s/^/[$tag] /"
}
run default VLM_GEMV_MFMA_DEBUG=1
run nosplit VLM_GEMV_MFMA_NORM_SPLIT=0
run seg28 VLM_GEMV_MFMA_SEG_CHUNKS=28 VLM_GEMV_MFMA_LDS_KB=160
run seg28ns VLM_GEMV_MFMA_SEG_CHUNKS=28 VLM_GEMV_MFMA_LDS_KB=160 VLM_GEMV_MFMA_NORM_SPLIT=0
run seg20 VLM_GEMV_MFMA_SEG_CHUNKS=20 VLM_GEMV_MFMA_LDS_KB=160
run wg1 VLM_GEMV_MFMA_WGS_PER_CU=1
run wg3 VLM_GEMV_MFMA_WGS_PER_CU=3
env VLM_GEMV_MFMA_DEBUG=1 timeout 200 python scripts/mfma_shapes.py mistral 7b 2b --rows 8 > $O/shapes_rows8.txt 2>&1; grep -E "^==|layer" $O/shapes_rows8.txt
grep "gemv_mfma\]" $O/shapes_default.txt | sort -u | head -40
