#!/bin/bash
# Round-4 GPU session 10: the default bench line of the current tree (configs block included)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_gpu10
mkdir -p $O
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
tail -c 600 $O/bench_err.txt
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_gpu10/bench_line.json').read().strip().splitlines()[-1])
print('value',d['value'],'frac',d['roofline']['frac'],'vit',d['roofline_vit']['frac'],d['vision_images_per_s'])
for k in ('batch8_decode','batch16_decode','wide64_decode','continuous_batching'):
    print(k,{a:b for a,b in d.get(k,{}).items() if 'tps' in a or 'tokens_per_s' in a})
for k,v in d.get('configs',{}).items():
    print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('decode_us_per_token'))
P
