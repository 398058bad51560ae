#!/bin/bash
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_31; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_sampler_gpu.py tests/test_engine_gpu.py -q -m gpu -x > $O/pytest.out 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.out
(cd /tmp && rm -rf /tmp/p4 && timeout -k 15 300 rocprofv3 --kernel-trace --stats -d /tmp/p4 -o b -- python3 $R/scripts/batch_prof.py 16 64 > $O/batchprof.log 2>&1); echo "batchprof rc=$?"
python3 $R/scripts/prof_summary.py $(find /tmp/p4 -name "*.db" | head -1) $O/r06_batch16_kernel_stats.txt > /dev/null 2>&1
grep "^B=" $O/batchprof.log; grep "logprob_argmax_tail\|lse_partial" $O/r06_batch16_kernel_stats.txt
python3 scripts/batch_prof.py 16 64 | grep "^B="; python3 scripts/batch_prof.py 8 64 | grep "^B="
python3 bench.py --stage headline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['decode_us_per_token'])"
(cd /tmp && rm -rf /tmp/p1 && timeout -k 15 400 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o p -- python3 $R/bench.py --stage headline --steps 3 --warmup 1 > $O/prof.log 2>&1); echo "prof rc=$?"
python3 $R/scripts/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) $O/headline_stats.txt > /dev/null 2>&1
grep "logprob_argmax_tail\|lse_partial" $O/headline_stats.txt
