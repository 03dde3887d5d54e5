#!/bin/bash
# round 5, call O: the estimator's forward on two fp16 planes (three products): parity tests, then timing, two and three workgroups per CU
mkdir -p gpurun_out/r5o
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r5o/est_tests.log
echo "pytest rc $?" >> gpurun_out/r5o/est_tests.log
timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16\|Error\|error" > gpurun_out/r5o/time_fwd2.log
DFEPE_LIB_PATH=/root/repo/ab_libs/libdfepe_fwd3.so timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16\|Error\|error" > gpurun_out/r5o/time_fwd3.log
timeout 300 python scripts/estimator_time.py 12 2000 2>&1 | grep "split-bf16\|stock\|Error\|error" > gpurun_out/r5o/time_12x2000.log
tail -n 30 gpurun_out/r5o/*.log
