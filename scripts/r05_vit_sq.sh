#!/bin/bash
# Round 5: the SQ counters of the ViT call at 16 images on the final kernels (the round-4 review asked for the attention kernel's
# VALU : MFMA mix again after its rework).  Two --pmc passes (kernel-trace only), the recipe of scripts/r02_vit_gpu.sh.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_vit_sq; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d /tmp/vitpmc -o vit -- python3 $R/scripts/vit_prof.py 16 > $O/vitpmc.log 2>&1; echo "a rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES -d /tmp/vitpmc2 -o vit -- python3 $R/scripts/vit_prof.py 16 > $O/vitpmc2.log 2>&1; echo "b rc=$?"
cd $R
python3 scripts/sq_pmc_summary.py $(find /tmp/vitpmc -name "*.db" | head -1) $O/r05_vit16_sq_pmc_a.txt > /dev/null
python3 scripts/sq_pmc_summary.py $(find /tmp/vitpmc2 -name "*.db" | head -1) $O/r05_vit16_sq_pmc_b.txt > /dev/null
grep -A9 attn_prefill $O/r05_vit16_sq_pmc_a.txt | head -10; grep -A8 attn_prefill $O/r05_vit16_sq_pmc_b.txt | head -9; grep -A8 "gemm256_kernel<3" $O/r05_vit16_sq_pmc_b.txt | head -9
