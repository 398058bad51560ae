#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call6; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_rotating_gpu.py tests/test_cache_contract_gpu.py tests/test_bench_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "rotating or reservation or phi3v_refuses or cache_facades or update_and_fetch or prefill_step_size or driver_command or headline_configuration or window" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -30 $O/pytest.out
