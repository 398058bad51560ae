#!/bin/bash
# ablations of the free-running GEMM (library built with VLM_BUILD_DEFINES=VLM_GEMM_ABLATION) + SQ counters of modes 3 and 16
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_02; mkdir -p $O
for shape in "8192 8192 8192" "36864 5120 1280"; do
  for abl in 0 1 2 3 4; do
    VLM_GEMM_F_ABL=$abl timeout 120 python3 scripts/r06/gemm_one.py $shape 16 2>/dev/null | tail -1 >> $O/abl.out
  done
  timeout 120 python3 scripts/r06/gemm_one.py $shape 3 2>/dev/null | tail -1 >> $O/abl.out
done
cat $O/abl.out
cd /tmp
for mode in 3 16; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d /tmp/pa$mode -o g -- python3 $R/scripts/r06/gemm_one.py 8192 8192 8192 $mode 5 > $O/pmc_a$mode.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES -d /tmp/pb$mode -o g -- python3 $R/scripts/r06/gemm_one.py 8192 8192 8192 $mode 5 > $O/pmc_b$mode.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d /tmp/pc$mode -o g -- python3 $R/scripts/r06/gemm_one.py 8192 8192 8192 $mode 5 > $O/pmc_c$mode.log 2>&1
  for x in a b c; do python3 $R/scripts/sq_pmc_summary.py $(find /tmp/p$x$mode -name "*.db" | head -1) $O/pmc_${x}_mode$mode.txt > /dev/null 2>>$O/err.log; done
done
cd $R
for f in $O/pmc_*_mode*.txt; do echo "== $f"; grep -A10 "gemm256" $f | head -12; done
