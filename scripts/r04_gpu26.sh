#!/bin/bash
# Session 26: max_kv_size (rotating window) GPU tests + the older tests around the module-contract decode path
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu26
mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_rotating_gpu.py -q -m gpu --tb=short -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -60 ) > $O/t_rot.log 2>&1; tail -40 $O/t_rot.log
( timeout 600 python -m pytest tests/test_parity_decode_gpu.py tests/test_kv_quant_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "tiny or teacher or peaked or quant" 2>&1 | tail -8 ) > $O/t_old.log 2>&1; tail -5 $O/t_old.log
