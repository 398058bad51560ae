#!/bin/bash
# Session 28: the payload-carrying typical-p sort - sampler tests first; only if they pass, the two --pmc passes on this tree's
# decode sources (sample.hip is one of them) and the per-filter kernel times.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu28
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_sampler_gpu.py tests/test_ops_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "sampl or filter or golden or xtc or typical or draw or batch_generator" -x > $O/t.log 2>&1
rc=$?
tail -6 $O/t.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc: no PMC passes"; exit 1; fi
SHORT="python $R/bench.py --steps 1 --warmup 0 --max-tokens 12 --no-cpu-baseline --no-extras"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $SHORT > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- $SHORT > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python scripts/pmc_summary.py $O/r04_pmc_traffic.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) | grep "gemv_rowwave_kernel<4, 3, 1, 1" | head -3
rm -rf $O/pmc_fetch $O/pmc_write
sed -i 's/r04_gpu25/r04_gpu28/g' scripts/r04_gpu25.sh
bash scripts/r04_gpu25.sh 2>&1 | grep -E "typical|top_p |classic|rc="
