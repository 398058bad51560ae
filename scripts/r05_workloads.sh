#!/bin/bash
# Round 5: the dedicated line of every other BASELINE config on the final .so (`bench.py --workload <name>`, its own extras and CPU
# baseline), one process each.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_workloads; mkdir -p $O
for w in nanollava qwen2vl-7b-b32 qwen2vl-2b-w4 idefics2-b8 phi35v-w4-b16; do
  S=$(date +%s)
  timeout 420 python3 bench.py --workload $w --steps 5 --warmup 2 > $O/r05_bench_$w.json 2> $O/$w.err; echo "$w rc=$? wall $(( $(date +%s) - S )) s"
  python3 - $O/r05_bench_$w.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", d["metric"][:60], round(d["value"], 1), d["unit"], "frac", round(d["roofline"]["frac"], 4), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("    NO LINE", e)
P
done
S=$(date +%s); timeout 300 python3 bench.py --workload phi35v-w4-b16 --kv-bits 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/r05_bench_phi35v-w4-b16_kv8.json 2> $O/phi_kv8.err; echo "phi kv8 rc=$? wall $(( $(date +%s) - S )) s"
