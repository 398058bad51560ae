#!/bin/bash
# Round-4 GPU session 18: ViT attention with the short query blocks dispatched last - bit-identity test + A/B (16 x 336^2 images)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu18
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "attn_prefill" 2>&1 | tail -3
for rep in 1 2; do for v in 1 0; do
  echo "== VLM_ATTN_SHORT_LAST=$v"
  VLM_ATTN_SHORT_LAST=$v timeout 120 python scripts/vit_prof.py 16 2>&1 | tail -1
done; done
for v in 1 0; do
  (cd /tmp && rm -rf /tmp/prof_v$v && VLM_ATTN_SHORT_LAST=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_v$v -o p -- python $R/scripts/vit_prof.py 16 > /tmp/prof_v$v.log 2>&1)
  db=$(find /tmp/prof_v$v -name "*.db" | head -1)
  python scripts/prof_summary.py $db $O/vit16_short_last_$v.txt > /dev/null 2>&1
  echo "== rocprof VLM_ATTN_SHORT_LAST=$v"; grep -E "attn_prefill|gemm256|layernorm" $O/vit16_short_last_$v.txt | cut -c1-140
done
