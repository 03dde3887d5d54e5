#!/bin/bash
# round 5, call Q: the data gradient fused with the adjoint below it: tests, then A/B of the estimator call and the full model
mkdir -p gpurun_out/r5q
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r5q/est_tests.log
timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16\|Error\|error" > gpurun_out/r5q/time_fused.log
DFEPE_EST_FUSE_DGRAD=0 timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16\|Error\|error" > gpurun_out/r5q/time_separate.log
timeout 600 python scripts/full_model_time.py > gpurun_out/r5q/full_model.log 2>&1
tail -n 16 gpurun_out/r5q/*.log
