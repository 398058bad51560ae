#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r06_18
python scripts/r06/sample_time.py 2>&1 | tail -2 > gpurun_out/r06_18/split.txt
VLM_SAMPLE_SPLIT=0 python scripts/r06/sample_time.py 2>&1 | tail -2 > gpurun_out/r06_18/single.txt
cat gpurun_out/r06_18/*.txt
