#!/bin/bash
# Round-4 GPU session 6: gemv_mfma2 - in-kernel norm correctness (ops tests) + standalone probes with phase stamps
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu6
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "mfma" 2>&1 | tail -4
B=scripts/bin
{
for nk in 1 0; do
  export VLM_GEMV_MFMA2_NORM_IN_KERNEL=$nk
  echo "== norm in kernel: $nk"
  $B/mfma2_probe_w4 16384 3072 16 16 1; $B/mfma2_probe_w4_st 16384 3072 16 16 1 | tail -1
  $B/mfma2_probe_w4 9216 3072 16 0 1;  $B/mfma2_probe_w4_st 9216 3072 16 0 1 | tail -1
  $B/mfma2_probe 17920 1536 16 16 1;   $B/mfma2_probe_st 17920 1536 16 16 1 | tail -1
  $B/mfma2_probe 2048 1536 16 0 1;     $B/mfma2_probe_st 2048 1536 16 0 1 | tail -1
done
unset VLM_GEMV_MFMA2_NORM_IN_KERNEL
echo "== no norm"
$B/mfma2_probe_w4 16384 3072 16 16 0; $B/mfma2_probe_w4_st 16384 3072 16 16 0 | tail -1
$B/mfma2_probe_w4 3072 8192 16 8 0;  $B/mfma2_probe_w4_st 3072 8192 16 8 0 | tail -1
$B/mfma2_probe_w4 3072 3072 16 8 0;  $B/mfma2_probe_w4_st 3072 3072 16 8 0 | tail -1
$B/mfma2_probe 17920 1536 16 16 0;   $B/mfma2_probe_st 17920 1536 16 16 0 | tail -1
$B/mfma2_probe 1536 8960 16 8 0;     $B/mfma2_probe_st 1536 8960 16 8 0 | tail -1
$B/mfma2_probe 1536 1536 16 8 0;     $B/mfma2_probe_st 1536 1536 16 8 0 | tail -1
$B/mfma2_probe 37888 3584 16 16 0;   $B/mfma2_probe_st 37888 3584 16 16 0 | tail -1
for w in 1 2 3; do echo "== WGS_PER_CU=$w"; VLM_GEMV_MFMA2_WGS_PER_CU=$w $B/mfma2_probe_w4 16384 3072 16 16 0; VLM_GEMV_MFMA2_WGS_PER_CU=$w $B/mfma2_probe 17920 1536 16 16 0; done
} > $O/probe.txt 2>&1
cat $O/probe.txt
