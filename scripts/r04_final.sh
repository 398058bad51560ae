#!/bin/bash
# Round-4 end-of-round evidence run on the GPU box (one gpurun call): the whole GPU suite, smoke, the two --pmc passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) behind roofline.traffic, rocprofv3 kernel summaries (default
# bench, 16-row step, ViT) and the default bench line.  Everything lands in gpurun_out/r04_final/; the summaries worth keeping
# are copied into profiles/ afterwards (the PMC file right away, so that the bench line of this run carries the traffic).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -q -m gpu --tb=line -p no:cacheprovider 2>&1 | grep -v "^$" | tail -15 ) > $O/t_all.log 2>&1; tail -6 $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
SHORT="python $R/bench.py --steps 1 --warmup 0 --max-tokens 12 --no-cpu-baseline --no-extras"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $SHORT > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- $SHORT > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python scripts/pmc_summary.py $O/r04_pmc_traffic.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) | head -8
cp $O/r04_pmc_traffic.json $R/profiles/r04_pmc_traffic.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_bench.log 2>&1; echo "prof rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_b16 -o s -- python $R/scripts/batch_prof.py 16 64 > $O/prof_b16.log 2>&1; echo "prof16 rc=$?"
timeout 250 rocprofv3 --kernel-trace --stats -d $O/prof_vit -o v -- python $R/scripts/vit_prof.py 16 > $O/prof_vit.log 2>&1; echo "vitprof rc=$?"
cd $R
python scripts/prof_summary.py $(find $O/prof_bench -name "*.db" | head -1) $O/r04_bench_kernel_stats.txt | head -12
python scripts/prof_summary.py $(find $O/prof_b16 -name "*.db" | head -1) $O/r04_batch16_kernel_stats.txt | head -10
python scripts/prof_summary.py $(find $O/prof_vit -name "*.db" | head -1) $O/r04_vit16_kernel_stats.txt | head -8
rm -rf $O/pmc_fetch $O/pmc_write $O/prof_bench $O/prof_b16 $O/prof_vit
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04_final/r04_bench_line.json').read().strip().splitlines()[-1])
print('value',d['value'],'frac',d['roofline']['frac'],'traffic',d['roofline'].get('traffic'),'vit',d['roofline_vit']['frac'])
for k in ('batch8_decode','batch16_decode','wide64_decode'):
    print(k,{a:round(b,1) for a,b in d.get(k,{}).items() if 'tps' in a})
for k,v in d.get('configs',{}).items():
    print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('error'))
P
