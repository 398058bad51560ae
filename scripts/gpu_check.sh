#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof.  Everything is logged under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi" > gpurun_out/env.log; rocm-smi --showproductname >> gpurun_out/env.log 2>&1; nproc >> gpurun_out/env.log
# health check first: a broken box must cost seconds, not the whole timeout budget
if ! timeout 120 python -c "import torch; x=torch.ones(1024,device='cuda'); assert float((x*2).sum().cpu())==2048.0; print('gpu ok', torch.cuda.get_device_name(0))" >> gpurun_out/env.log 2>&1; then
  echo "GPU HEALTH CHECK FAILED - aborting"; tail -5 gpurun_out/env.log; exit 3
fi
STEPS="${1:-ops engine smoke bench prof}"
for s in $STEPS; do
  case $s in
    ops)    timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/test_ops.log 2>&1; echo "ops rc=$?" ;;
    engine) timeout 300 python -m pytest tests/test_engine_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/test_engine.log 2>&1; echo "engine rc=$?" ;;
    smoke)  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" ;;
    bench)  timeout 420 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench.log ;;
    prof)   (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1); echo "prof rc=$?" ;;
  esac
done
for f in test_ops test_engine smoke; do [ -f gpurun_out/$f.log ] && { echo "---- $f"; tail -n 25 gpurun_out/$f.log; }; done
