#!/bin/bash
# round 2, one GPU visit for the vision tower: A/B of the fused rope epilogue, kernel trace, SQ counters of the attention
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== ViT 16 x 336^2, rope in the qkv epilogue" > $O/vit_r02.txt
timeout 200 python $R/scripts/vit_prof.py 16 >> $O/vit_r02.txt 2>&1
echo "== ViT 16 x 336^2, separate rope pass" >> $O/vit_r02.txt
VLM_VIT_ROPE_FUSED=0 timeout 200 python $R/scripts/vit_prof.py 16 >> $O/vit_r02.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/vitprof2 -o vit -- python $R/scripts/vit_prof.py 16 > $O/vitprof2.log 2>&1
DB=$(ls $O/vitprof2/*/*.db $O/vitprof2/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/prof_summary.py $DB $O/r02_vit16_kernel_stats.txt > /dev/null
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > $O/sq_counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/vitpmc -o vit -- python $R/scripts/vit_prof.py 16 > $O/vitpmc.log 2>&1
DB=$(ls $O/vitpmc/*/*.db $O/vitpmc/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/sq_pmc_summary.py $DB $O/r02_vit16_sq_pmc.txt > /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES -d $O/vitpmc2 -o vit -- python $R/scripts/vit_prof.py 16 > $O/vitpmc2.log 2>&1
DB=$(ls $O/vitpmc2/*/*.db $O/vitpmc2/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/sq_pmc_summary.py $DB $O/r02_vit16_sq_pmc2.txt > /dev/null
# keep the merge-back small: the rocpd databases stay on the box
rm -rf $O/vitprof2 $O/vitpmc $O/vitpmc2
cat $O/vit_r02.txt; head -14 $O/r02_vit16_kernel_stats.txt; grep -A9 attn_prefill $O/r02_vit16_sq_pmc.txt; grep -A8 attn_prefill $O/r02_vit16_sq_pmc2.txt
