#!/bin/bash
# Round-4 GPU session 8: fused decode block - layout 1 (the chain on CUs of its own) with a paced weight stream: tests,
# timelines and the end-to-end A/B over the depth of the stream; layout 0 (first cut) timeline for the record
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu8
mkdir -p $O
cd $R
echo "== tests/test_decode_block_gpu.py (layout 1, depth 8)"
timeout 420 python -m pytest tests/test_decode_block_gpu.py -q -x 2>&1 | tail -6
for cfg in "1 4" "1 8" "1 16" "1 64" "0 64"; do
  set -- $cfg
  export VLM_DECODE_BLOCK_LAYOUT=$1 VLM_DECODE_BLOCK_DEPTH=$2
  echo "== layout $1 depth $2"
  timeout 120 python scripts/block_stamps.py 450 20 > $O/stamps_l$1_d$2.txt 2>&1; grep -v amdgpu.ids $O/stamps_l$1_d$2.txt | tail -8
  timeout 120 scripts/bin/decode_probe --steps 200 --ctx 450 --no-hot --variant 1,16,1,1,0 --variant 1,16,1,1,1 > $O/probe_l$1_d$2.txt 2>&1; grep "us/step\|GAVE" $O/probe_l$1_d$2.txt
done
unset VLM_DECODE_BLOCK_LAYOUT VLM_DECODE_BLOCK_DEPTH
