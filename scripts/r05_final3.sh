#!/bin/bash
# Round-5 evidence run, third part: after the sampled top-p step lost two launches (log-probs inside the histogram launch, the
# Gumbel draw inside the mask launch - csrc/sample.hip, which is part of the decode-source hash of the PMC file).  Greedy decode and
# ViT kernels are untouched.  Order: what the bench line needs first (PMC passes on the final sources, the driver's command, one
# short repeat), then the whole GPU suite.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_final
mkdir -p $O
SHORT="python3 $R/bench.py --stage headline --gpus 1 --steps 1 --warmup 0 --max-tokens 12"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $SHORT > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- $SHORT > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python3 scripts/pmc_summary.py $O/r05_pmc_traffic.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) | grep -E "gemv_rowwave_kernel<4, 3, 1, 1, 16>|splitk"
cp $O/r05_pmc_traffic.json $R/profiles/r05_pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
S=$(date +%s); timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_line_1.json 2> $O/bench_1.err; echo "bench 1 rc=$? wall $(( $(date +%s) - S )) s"
S=$(date +%s); timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/r05_bench_line_2.json 2> $O/bench_2.err; echo "bench 2 rc=$? wall $(( $(date +%s) - S )) s"
python3 - <<'P'
import json
for i in (1,2):
    try:
        d=json.loads(open(f'gpurun_out/r05_final/r05_bench_line_{i}.json').read().strip().splitlines()[-1])
    except Exception as e:
        print(i,'NO LINE',e); continue
    print(i,'value',round(d['value'],1),'frac',round(d['roofline']['frac'],4),'traffic',d['roofline'].get('traffic'),d['roofline'].get('traffic_source'),'vit',round(d['roofline_vit']['frac'],4),
          'attempts',d.get('headline_attempts'),'nan_rows',d.get('decode_nan_rows'),'cpu',d.get('cpu_baseline',{}).get('value'))
    print('  sampled', {k: round(v['decode_us_per_token'],1) for k,v in (d.get('sampled_decode') or {}).items()})
    if i==1:
        for k,v in (d.get('configs') or {}).items():
            print(' ',k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('error'))
P
( time timeout 1800 python3 -m pytest tests -v -m gpu --tb=line -p no:cacheprovider 2>&1 | grep -v "^$" ) > $O/t_all_verbose.log 2>&1
tail -15 $O/t_all_verbose.log > $O/t_all.log; tail -6 $O/t_all.log
grep -E "test_rotating_gpu|test_cache_contract_gpu|test_bench_gpu|test_sampler_gpu.*split|stop_token_and_remove| passed| failed" $O/t_all_verbose.log > $O/r05_rotating_cache_bench_gpu_tests.txt; tail -1 $O/r05_rotating_cache_bench_gpu_tests.txt
