#!/bin/bash
# round 5, call AE: estimator + captured-step + compat tests with the small-grid GEMM builds (two tiles ahead, split stage)
mkdir -p gpurun_out/r5ae
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py tests/test_captured_step_gpu.py tests/test_compat_gpu.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r5ae/tests.log
DFEPE_EST_SMALL_GRID=0 timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py -q -m gpu 2>&1 | tail -3 >> gpurun_out/r5ae/tests.log
cat gpurun_out/r5ae/tests.log
