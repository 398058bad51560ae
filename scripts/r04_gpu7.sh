#!/bin/bash
# Round-4 GPU session 7: the fused decode block (csrc/decode_block.hip) - bit-identity tests, timeline, end-to-end A/B
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu7
mkdir -p $O
cd $R
echo "== tests/test_decode_block_gpu.py"
timeout 420 python -m pytest tests/test_decode_block_gpu.py -q -x 2>&1 | tail -15
echo "== block_stamps ctx 450"
timeout 180 python scripts/block_stamps.py 450 20 > $O/block_stamps_450.txt 2>&1; grep -v amdgpu.ids $O/block_stamps_450.txt | tail -12
echo "== block_stamps ctx 1000"
timeout 180 python scripts/block_stamps.py 1000 20 > $O/block_stamps_1000.txt 2>&1; grep -v amdgpu.ids $O/block_stamps_1000.txt | tail -12
echo "== decode_probe"
timeout 240 scripts/bin/decode_probe --steps 300 --ctx 450 --no-hot > $O/decode_probe.txt 2>&1; cat $O/decode_probe.txt | tail -12
echo "== decode_probe, block table"
timeout 240 scripts/bin/decode_probe --steps 300 --ctx 450 --no-hot --block-table > $O/decode_probe_bt.txt 2>&1; cat $O/decode_probe_bt.txt | tail -8
echo "== engine tests (fused block on by default)"
timeout 900 python -m pytest tests/test_parity_decode_gpu.py tests/test_engine_gpu.py -q -x 2>&1 | tail -8
