#!/bin/bash
# round 5, call N: timing probe -- the estimator's forward with three products on two planes (what a two-plane fp16 forward would issue)
mkdir -p gpurun_out/r5n
cd /root/repo
timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16" > gpurun_out/r5n/six_products.log
DFEPE_EST_PROBE_2P=1 timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16" > gpurun_out/r5n/three_products.log
cat gpurun_out/r5n/*.log
