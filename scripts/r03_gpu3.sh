#!/bin/bash
# Round-3 GPU session 3: correctness after the permlane fix + new features (per-request processors, batched Su-RoPE regime),
# full-depth parity numbers, whole suite with the complete failure list.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_s3
mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_full_depth_gpu.py --tb=line 2>&1 | grep -v "^$" | tail -150 ) > $O/t_all.log 2>&1; tail -60 $O/t_all.log
( time timeout 1500 python -m pytest tests/test_full_depth_gpu.py -q -s --tb=line 2>&1 | grep -E "rel-rms|passed|failed|FAILED|Error|error" | tail -40 ) > $O/t_full.log 2>&1; cat $O/t_full.log
P=scripts/bin/decode_probe
V="--variant 0,256,0x7f,1,96"
timeout 300 $P --steps 300 --ctx 450 --no-hot $V,0,0,0,0,1 $V,0,0,16,1,1 $V,0,0,16,1,0 > $O/probe_ctx450.txt 2>&1; cat $O/probe_ctx450.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json,sys
d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r03_s3/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","decode_us_per_token")}, d["roofline"]["frac"], d.get("roofline_vit",{}).get("frac"), d.get("batch16_decode"), d.get("batch8_decode"))
PY
