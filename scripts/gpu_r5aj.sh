#!/bin/bash
# round 5, call AJ: one library call per estimator pass: bit-identity tests, the whole estimator suite on it, small-batch timing A/B
mkdir -p gpurun_out/r5aj
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py tests/test_captured_step_gpu.py tests/test_compat_gpu.py tests/test_estimator_gpu.py tests/test_api_path_gpu.py -q -m gpu 2>&1 | tail -12 > gpurun_out/r5aj/tests.log
cat gpurun_out/r5aj/tests.log gpurun_out/r5aj/small.log
