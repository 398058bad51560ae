#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_27; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_full_depth_gpu.py tests/test_engine_gpu.py -q -m gpu -x -k "wide" > $O/pytest.out 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest.out
for t in 0 1 0 1; do VLM_GEMM_SKINNY64=$t VLM_WIDE_TAILS=$t python3 scripts/r06/wide_prof.py 64 2>/dev/null | tail -1 | cut -c1-200; done
