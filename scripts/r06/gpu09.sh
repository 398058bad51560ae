#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_09; mkdir -p $O
for h in 0 1 0 1; do
  VLM_ATTN_PREFILL_HILO=$h timeout 300 python3 scripts/r05_attn_bench.py 2>/dev/null | sed "s/^/hilo=$h /" >> $O/attn_bench.out
done
VLM_ATTN_PREFILL_HILO=1 timeout 600 python3 -m pytest tests/test_op_noise_gpu.py -q -m gpu -s -k "prefill_flash" 2>&1 | grep -E "HIP vs exact|passed|failed" | cut -c1-150 > $O/noise_hilo.out
VLM_ATTN_PREFILL_HILO=1 timeout 900 python3 -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "attn_prefill" 2>&1 | tail -3 >> $O/noise_hilo.out
cat $O/attn_bench.out $O/noise_hilo.out
