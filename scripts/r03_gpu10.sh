#!/bin/bash
# Session 10: deferred hand-off of the K-split forms (tickets + merges after a workgroup's last unit) - tests, then the
# per-projection times with / without it and under the K-segment policies.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s10
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "gemv_mfma or gemv_w4_mfma" 2>&1 | tail -15 ) > $O/t_mfma.log 2>&1; tail -4 $O/t_mfma.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python scripts/mfma_shapes.py 2b 7b mistral phi-w4 --rows 16 > $O/shapes_$tag.txt 2>&1
  grep -E "^==|layer|qkv norm\+rope|qkv norm\+bias|o_proj|gate|down" $O/shapes_$tag.txt | awk '/^==/ {printf "%s:", $0; next} /layer/ {printf " layer %s us %s TB/s\n", $5, $7; next} {printf " %s %s |", $1, $(NF-3)}' | sed "s/^/[$tag] /"
}
run inline VLM_GEMV_MFMA_DEFER=0
run defer VLM_GEMV_MFMA_DEFER=1
run defer_ns VLM_GEMV_MFMA_DEFER=1 VLM_GEMV_MFMA_NORM_SPLIT=0
run defer_seg28 VLM_GEMV_MFMA_DEFER=1 VLM_GEMV_MFMA_SEG_CHUNKS=28 VLM_GEMV_MFMA_LDS_KB=160
run defer_seg8 VLM_GEMV_MFMA_DEFER=1 VLM_GEMV_MFMA_SEG_CHUNKS=8
run defer_seg6_wg3 VLM_GEMV_MFMA_DEFER=1 VLM_GEMV_MFMA_SEG_CHUNKS=6 VLM_GEMV_MFMA_WGS_PER_CU=3
run defer_split2k VLM_GEMV_MFMA_DEFER=1 VLM_GEMV_MFMA_NORM_SPLIT=2048
