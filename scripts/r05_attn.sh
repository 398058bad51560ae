#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_attn2; mkdir -p $O
for p in 0 1; do
  VLM_ATTN_PIPE=$p timeout 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attn_prefill" > $O/pytest_ops$p.out 2>&1; echo "pytest_ops pipe=$p rc=$?" >> $O/rc.txt
  VLM_ATTN_PIPE=$p timeout 300 python3 scripts/r05_attn_bench.py > $O/bench$p.out 2>&1
done
VLM_ATTN_PIPE=0 timeout 600 python3 -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "tower or vit or vision or features" > $O/pytest_tower.out 2>&1; echo "pytest_tower rc=$?" >> $O/rc.txt
for p in 0 1; do VLM_ATTN_PIPE=$p timeout 300 python3 scripts/r05_vit_sweep.py 16 64 > $O/sweep$p.out 2>&1; done
cat $O/rc.txt; tail -2 $O/pytest_ops0.out; tail -2 $O/pytest_ops1.out; tail -2 $O/pytest_tower.out; grep pipe $O/bench0.out $O/bench1.out; tail -1 $O/sweep0.out; tail -1 $O/sweep1.out
