#!/bin/bash
# Round-4 GPU session 2: the attention-block skeleton probe (what ONE launch for qkv -> attention -> o_proj would
# cost, before building it) and the default bench line with its new `configs` block (item 4), timed.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu2
mkdir -p $O
cd $R
timeout 120 scripts/bin/attn_block_probe > $O/attn_block_probe.txt 2>&1; echo "probe rc=$?"; tail -40 $O/attn_block_probe.txt
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
tail -c 6000 $O/bench_default.json; echo; tail -5 $O/bench_default.err
