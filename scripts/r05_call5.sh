#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_call5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_cache_contract_gpu.py tests/test_engine_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "cache_facades or update_and_fetch or prefill_step_size or rotating or max_kv or abi or symbols or nan_row" > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
ls tests | grep -i rotat > $O/ls.txt
timeout 900 python3 -m pytest tests -x -q -m gpu -k "rotating or max_kv_size" --ignore=tests/test_full_depth_gpu.py > $O/pytest_rot.out 2>&1; echo "pytest_rot rc=$?" >> $O/rc.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-configs > $O/bench.out 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -4 $O/pytest.out; tail -4 $O/pytest_rot.out
python3 -c "
import json
d=json.loads(open('$O/bench.out').read().strip().splitlines()[-1])
print('value',d['value'],'frac',d['roofline']['frac'],'vit',d.get('roofline_vit'),'sweep',d.get('vision_batch_sweep_336'))
"
