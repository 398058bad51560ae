#!/bin/bash
# Round-5 evidence run, second part (same .so as scripts/r05_final.sh, whose bench lines were lost to a missing /usr/bin/time on the
# box and whose suite had one tie-sensitive test to fix): the whole GPU suite again, then the driver's command itself and two shorter
# repeats of its headline.  The PMC file of the first part is already in profiles/ (it travels with the snapshot).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
( time timeout 1800 python3 -m pytest tests -v -m gpu --tb=line -p no:cacheprovider 2>&1 | grep -v "^$" ) > $O/t_all_verbose.log 2>&1
tail -15 $O/t_all_verbose.log > $O/t_all.log; tail -6 $O/t_all.log
grep -E "test_rotating_gpu|test_cache_contract_gpu|test_bench_gpu|test_sampler_gpu.*split|stop_token_and_remove| passed| failed" $O/t_all_verbose.log > $O/r05_rotating_cache_bench_gpu_tests.txt; tail -1 $O/r05_rotating_cache_bench_gpu_tests.txt
S=$(date +%s); timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_line_1.json 2> $O/bench_1.err; echo "bench 1 rc=$? wall $(( $(date +%s) - S )) s"; tail -c 200 $O/bench_1.err
for i in 2 3; do
  S=$(date +%s); timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/r05_bench_line_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$? wall $(( $(date +%s) - S )) s"; tail -c 200 $O/bench_$i.err
done
python3 - <<'P'
import json
for i in (1,2,3):
    try:
        d=json.loads(open(f'gpurun_out/r05_final/r05_bench_line_{i}.json').read().strip().splitlines()[-1])
    except Exception as e:
        print(i,'NO LINE',e); continue
    print(i,'value',round(d['value'],1),'frac',round(d['roofline']['frac'],4),'traffic',d['roofline'].get('traffic'),'vit',round(d['roofline_vit']['frac'],4),
          'attempts',d.get('headline_attempts'),'nan_rows',d.get('decode_nan_rows'),'cpu',d.get('cpu_baseline',{}).get('value'),'clocks',d.get('gpu_clocks'))
    print('  sampled', d.get('sampled_decode'))
    if i==1:
        for k in ('batch8_decode','batch16_decode','wide64_decode'):
            print(' ',k,{a:round(b,1) for a,b in (d.get(k) or {}).items() if 'tps' in a})
        for k,v in (d.get('configs') or {}).items():
            print(' ',k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('gpu_clocks'), v.get('error'))
P
