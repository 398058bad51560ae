#!/bin/bash
cat > /tmp/b8.py <<'P'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
from mlx_vlm_amd import synthetic
from mlx_vlm_amd.models import qwen2_vl
dev = torch.device("cuda", 0)
cfg, model, _ = bench._load_synthetic(synthetic.QWEN2_VL_2B, qwen2_vl, 0, dev, kv_pool_tokens=16384, max_seqs=16)
r = bench.batch_decode_throughput(model, cfg, 8, 64)
print(r)
P
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o m --output-format csv -- python /tmp/b8.py > /tmp/m.log 2>&1
tail -2 /tmp/m.log | cut -c1-200
python - <<'P'
import csv,re
rows=list(csv.reader(open("/tmp/prof_m/m_kernel_stats.csv")))
for r in rows[1:14]:
    m=re.search(r"(\w+)<([^>]{0,40})",r[0]); print((m.group(1)+"<"+m.group(2)+">") if m else r[0][:50], r[1], r[3], r[5], r[6])
P
