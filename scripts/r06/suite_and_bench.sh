#!/bin/bash
# full GPU suite + the driver's bench command on the current tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final; mkdir -p $O
timeout 3000 python3 -m pytest tests -q -m gpu -x > $O/pytest.out 2>&1; echo "pytest rc=$?" >> $O/rc.txt
timeout 1500 python3 bench.py --gpus 1 --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -4 $O/pytest.out; python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06_final/bench.json"))
print("value", d["value"], "us/token", d["decode_us_per_token"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print("vit", d["roofline"].get("vit"), "\nvit16", d["roofline"].get("vit_16"), "\nvit448", d["roofline"].get("vit_single_448"))
print("config", {k: d["config"][k] for k in ("headline_retries", "vit_batch")})
for k in ("batch8_decode","batch16_decode","wide64_decode","sampled_decode"): print(k, json.dumps(d.get(k))[:400])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
print("configs", json.dumps(d.get("configs"))[:1500])
PY
