#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06_17
timeout 900 python -m pytest tests/test_sampler_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_17/sampler.txt
cat gpurun_out/r06_17/sampler.txt
timeout 600 python scripts/r06/sampled_bench.py 2>&1 | tail -3 > gpurun_out/r06_17/sampled.txt
cat gpurun_out/r06_17/sampled.txt
