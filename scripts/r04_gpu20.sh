#!/bin/bash
# Round-4 GPU session 20: single-stream step, page-split width / GEMV variant sweep at ctx 450 and 1500 (decode_probe)
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_gpu20
for ctx in 450 1500; do
  echo "== ctx $ctx"
  timeout 200 scripts/bin/decode_probe --steps 300 --ctx $ctx --no-hot --variant 1,16,1,1 --variant 1,8,1,1 --variant 1,4,1,1 --variant 1,32,1,1 --variant 1,16,0,1 --variant 1,16,2,1 --variant 1,16,1,1 2>&1 | grep "us/step" | tee -a gpurun_out/r04_gpu20/sweep.txt
done
