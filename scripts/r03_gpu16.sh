#!/bin/bash
# Session 16: SQ counters of the 16-row decode step (separate --pmc passes, kernel-trace only) - what the batched decode GEMM
# spends its wave cycles on.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s16
mkdir -p $O
cd /tmp
timeout 250 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/pmc1 -o s -- python $R/scripts/batch_prof.py 16 24 > $O/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 250 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES -d $O/pmc2 -o s -- python $R/scripts/batch_prof.py 16 24 > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"
cd $R
python scripts/sq_pmc_summary.py $(find $O/pmc1 -name "*.db" | head -1) $O/r03_batch16_sq_pmc_a.txt > /dev/null
python scripts/sq_pmc_summary.py $(find $O/pmc2 -name "*.db" | head -1) $O/r03_batch16_sq_pmc_b.txt > /dev/null
rm -rf $O/pmc1 $O/pmc2
grep -A9 "gemv_mfma_kernel<1, 16" $O/r03_batch16_sq_pmc_a.txt | head -12; grep -A9 "gemv_mfma_kernel<1, 16" $O/r03_batch16_sq_pmc_b.txt | head -12
grep -A9 "gemv_mfma_kernel<0, 8, false" $O/r03_batch16_sq_pmc_a.txt | head -12; grep -A9 "gemv_mfma_kernel<0, 8, false" $O/r03_batch16_sq_pmc_b.txt | head -12
