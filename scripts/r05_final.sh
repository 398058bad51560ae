#!/bin/bash
# Round-5 end-of-round evidence run on the GPU box (ONE gpurun call, and the LAST one of the round - VERDICT r04: the final .so must
# not ship unbenched): the whole GPU suite (incl. tests/test_bench_gpu.py = the driver's command), smoke, the two --pmc passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, kernel-trace only) behind roofline.traffic, rocprofv3 kernel summaries (headline stage,
# ViT at 16 and 64 images, the 4-bit model at one row), then the driver's command itself and two shorter repeats of its headline.  Everything lands in gpurun_out/r05_final/; the
# summaries worth keeping are copied into profiles/ afterwards (the PMC file right away, so that the bench lines of this run
# carry the traffic).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
( time timeout 1800 python3 -m pytest tests -v -m gpu --tb=line -p no:cacheprovider 2>&1 | grep -v "^$" ) > $O/t_all_verbose.log 2>&1
tail -15 $O/t_all_verbose.log > $O/t_all.log; tail -6 $O/t_all.log
grep -E "test_rotating_gpu|test_cache_contract_gpu|test_bench_gpu|test_sampler_gpu.*split| passed| failed" $O/t_all_verbose.log > $O/r05_rotating_cache_bench_gpu_tests.txt; tail -1 $O/r05_rotating_cache_bench_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
SHORT="python3 $R/bench.py --stage headline --gpus 1 --steps 1 --warmup 0 --max-tokens 12"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $SHORT > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- $SHORT > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python3 scripts/pmc_summary.py $O/r05_pmc_traffic.json $(find $O/pmc_fetch -name "*.db" | head -1) $(find $O/pmc_write -name "*.db" | head -1) | head -8
cp $O/r05_pmc_traffic.json $R/profiles/r05_pmc_traffic.json
# the driver's command, as the driver runs it (line 1: everything) - before the profiles, so that a long run cannot cost it
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_line_1.json 2> $O/bench_1.err; echo "bench 1 rc=$?"; tail -c 200 $O/bench_1.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python3 $R/bench.py --stage headline --gpus 1 --steps 2 --warmup 1 > $O/prof_bench.log 2>&1; echo "prof rc=$?"
timeout 250 rocprofv3 --kernel-trace --stats -d $O/prof_vit -o v -- python3 $R/scripts/vit_prof.py 16 > $O/prof_vit.log 2>&1; echo "vitprof16 rc=$?"
timeout 250 rocprofv3 --kernel-trace --stats -d $O/prof_vit64 -o v -- python3 $R/scripts/vit_prof.py 64 > $O/prof_vit64.log 2>&1; echo "vitprof64 rc=$?"
# the 4-bit language model at ONE row, launch by launch, beside the bf16 step's launches
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_w4 -o q -- python3 $R/bench.py --workload qwen2vl-2b-w4 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/prof_w4.log 2>&1; echo "w4prof rc=$?"
cd $R
python3 scripts/prof_summary.py $(find $O/prof_bench -name "*.db" | head -1) $O/r05_bench_kernel_stats.txt | head -10
python3 scripts/r05_decode_gaps.py $(find $O/prof_bench -name "*.db" | head -1) $O/r05_decode_launch_durations.txt | head -12
python3 scripts/prof_summary.py $(find $O/prof_vit -name "*.db" | head -1) $O/r05_vit16_kernel_stats.txt | head -8
python3 scripts/prof_summary.py $(find $O/prof_vit64 -name "*.db" | head -1) $O/r05_vit64_kernel_stats.txt | head -8
python3 scripts/prof_summary.py $(find $O/prof_w4 -name "*.db" | head -1) $O/r05_w4_onerow_kernel_stats.txt | head -10
rm -rf $O/pmc_fetch $O/pmc_write $O/prof_bench $O/prof_vit $O/prof_vit64 $O/prof_w4
# the same headline + extras twice more without the CPU legs and the other configs (lines 2, 3: run-to-run spread)
for i in 2 3; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/r05_bench_line_$i.json 2> $O/bench_$i.err; echo "bench $i rc=$?"; tail -c 200 $O/bench_$i.err
done
python3 - <<'P'
import json
for i in (1,2,3):
    try:
        d=json.loads(open(f'gpurun_out/r05_final/r05_bench_line_{i}.json').read().strip().splitlines()[-1])
    except Exception as e:
        print(i,'NO LINE',e); continue
    print(i,'value',round(d['value'],1),'frac',round(d['roofline']['frac'],4),'traffic',d['roofline'].get('traffic'),'vit',round(d['roofline_vit']['frac'],4),
          'attempts',d.get('headline_attempts'),'nan_rows',d.get('decode_nan_rows'),'cpu',d.get('cpu_baseline',{}).get('value'))
    if i==1:
        for k in ('batch8_decode','batch16_decode','wide64_decode'):
            print(k,{a:round(b,1) for a,b in (d.get(k) or {}).items() if 'tps' in a})
        for k,v in (d.get('configs') or {}).items():
            print(k, v.get('value'), v.get('roofline',{}).get('frac'), v.get('error'))
P
