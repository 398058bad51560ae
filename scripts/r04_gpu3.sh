#!/bin/bash
# Round-4 GPU session 3: image-parallel ViT launch chains - bit identity test + A/B timing (1..4 chains, 8 / 16 / 32 images)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu3
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_engine_gpu.py -q -x -k "vision_tower or user_api" 2>&1 | tail -5
timeout 300 python scripts/vit_streams_ab.py > $O/vit_streams_ab.txt 2>&1; echo "ab rc=$?"; cat $O/vit_streams_ab.txt | grep -v amdgpu.ids
