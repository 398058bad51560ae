#!/bin/bash
# Round-4 GPU session 21: LayerNorm folded into the ViT's qkv / fc1 GEMMs - operator tests, tower tests, A/B (16 x 336^2 images)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu21
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "lnfold or ln_stats or gemm" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_parity_decode_gpu.py -q -x -k "vit or vision or image or full_size or baseline or prefill" 2>&1 | tail -5
for rep in 1 2; do for v in 1 0; do
  echo "== VLM_VIT_LNFOLD=$v"
  VLM_VIT_LNFOLD=$v timeout 120 python scripts/vit_prof.py 16 2>&1 | tail -1
done; done
(cd /tmp && rm -rf /tmp/prof_f && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o p -- python $R/scripts/vit_prof.py 16 > /tmp/prof_f.log 2>&1)
python scripts/prof_summary.py $(find /tmp/prof_f -name "*.db" | head -1) $O/vit16_lnfold.txt > /dev/null 2>&1
grep -E "attn_prefill|gemm256|layernorm|ln_stats" $O/vit16_lnfold.txt | cut -c1-140
