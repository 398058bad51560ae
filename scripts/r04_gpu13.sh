#!/bin/bash
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kv_quant_gpu.py -q -x -k "batch_generator_with_kv_bits" 2>&1 | tail -60
