#!/bin/bash
# Round-4 GPU session 14: the whole GPU suite on the current tree (no -x: every failure listed)
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_gpu14
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/r04_gpu14/pytest_tail.txt
