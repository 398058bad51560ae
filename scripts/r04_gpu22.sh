#!/bin/bash
# Round-4 GPU session 22: LN fold with the epilogue operands loaded at kernel entry - tests, A/B, profile
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu22
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_ops_gpu.py -q -s -k "lnfold or ln_stats" 2>&1 | grep -E "lnfold|passed|failed|Error|assert" | head -12
for rep in 1 2; do for v in 1 0; do
  echo "== VLM_VIT_LNFOLD=$v"
  VLM_VIT_LNFOLD=$v timeout 120 python scripts/vit_prof.py 16 2>&1 | tail -1
done; done
(cd /tmp && rm -rf /tmp/prof_f && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o p -- python $R/scripts/vit_prof.py 16 > /tmp/prof_f.log 2>&1)
python scripts/prof_summary.py $(find /tmp/prof_f -name "*.db" | head -1) $O/vit16_lnfold.txt > /dev/null 2>&1
grep -E "attn_prefill|gemm256|layernorm|ln_stats" $O/vit16_lnfold.txt | cut -c1-140
