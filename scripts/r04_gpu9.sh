#!/bin/bash
# Round-4 GPU session 9: fused decode block, layout 2 (the chain alone: attention + merge + o_proj in one launch of 32
# workgroups, gate/up as its own launch) - tests, timeline, end-to-end A/B
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu9
mkdir -p $O
cd $R
export VLM_DECODE_BLOCK_LAYOUT=2
echo "== tests/test_decode_block_gpu.py (layout 2)"
timeout 420 python -m pytest tests/test_decode_block_gpu.py -q -x 2>&1 | tail -6
timeout 120 python scripts/block_stamps.py 450 20 > $O/stamps_l2.txt 2>&1; grep -v amdgpu.ids $O/stamps_l2.txt | tail -8
timeout 120 scripts/bin/decode_probe --steps 300 --ctx 450 --no-hot --variant 1,16,1,1,0 --variant 1,16,1,1,1 --variant 1,16,1,1,0 --variant 1,16,1,1,1 > $O/probe_l2.txt 2>&1; grep "us/step\|GAVE" $O/probe_l2.txt
timeout 120 scripts/bin/decode_probe --steps 300 --ctx 1500 --no-hot --variant 1,16,1,1,0 --variant 1,16,1,1,1 > $O/probe_l2_ctx1500.txt 2>&1; grep "us/step\|GAVE" $O/probe_l2_ctx1500.txt
