#!/bin/bash
# Session 11: NSETS register sets of weights in flight (3 at 3 chunks per wave, 2 at 7) - tests, per-projection times under the
# K-segment policies, the 16-row 2B step and the 7B batch-32 job.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s11
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "gemv_mfma or gemv_w4_mfma" 2>&1 | tail -15 ) > $O/t_mfma.log 2>&1; tail -4 $O/t_mfma.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 200 python scripts/mfma_shapes.py 2b 7b mistral phi-w4 --rows 16 > $O/shapes_$tag.txt 2>&1
  grep -E "^==|layer|qkv norm\+rope|qkv norm\+bias|o_proj|gate|down" $O/shapes_$tag.txt | awk '/^==/ {printf "%s:", $0; next} /layer/ {printf " layer %s us %s TB/s\n", $5, $7; next} {printf " %s %s |", $1, $(NF-3)}' | sed "s/^/[$tag] /"
}
run sets
run sets_ns VLM_GEMV_MFMA_NORM_SPLIT=0
run sets_seg28 VLM_GEMV_MFMA_SEG_CHUNKS=28 VLM_GEMV_MFMA_LDS_KB=160 VLM_GEMV_MFMA_NORM_SPLIT=0
run sets_seg8 VLM_GEMV_MFMA_SEG_CHUNKS=8 VLM_GEMV_MFMA_NORM_SPLIT=0
run sets_wg3 VLM_GEMV_MFMA_WGS_PER_CU=3 VLM_GEMV_MFMA_NORM_SPLIT=0
timeout 300 python scripts/batch_prof.py 16 64 2>&1 | tail -1
timeout 300 python scripts/batch_prof.py 8 64 2>&1 | tail -1
