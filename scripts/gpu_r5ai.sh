#!/bin/bash
mkdir -p gpurun_out/r5ai; cd /root/repo
timeout 500 python scripts/stress_estimator.py 40 1 > gpurun_out/r5ai/stress.log 2>&1
tail -25 gpurun_out/r5ai/stress.log | cut -c1-200
