#!/bin/bash
# Round-3 GPU session 5: 8-bit KV cache (operators + engine), per-request processors, phi3 regime tests.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s5
mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_ops_gpu.py -q --tb=short -k "q8 or kv_quantize or paged_split or partial_only" 2>&1 | grep -v "^$" | tail -60 ) > $O/t_ops.log 2>&1; tail -40 $O/t_ops.log
( time timeout 600 python -m pytest tests/test_kv_quant_gpu.py -q -s --tb=short 2>&1 | grep -v "^$" | tail -80 ) > $O/t_kv.log 2>&1; tail -60 $O/t_kv.log
( time timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_vlm_family_phi3v_gpu.py -q --tb=line -k "per_request or su_rope or crossing or batch_generator" 2>&1 | grep -v "^$" | tail -30 ) > $O/t_misc.log 2>&1; tail -20 $O/t_misc.log
