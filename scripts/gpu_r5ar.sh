#!/bin/bash
# round 5, call AR: two LDS stages in the small-grid GEMM builds: tests, kernel-level A/B at B = 8, step-level timing
mkdir -p gpurun_out/r5ar
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py tests/test_captured_step_gpu.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r5ar/tests.log
cat gpurun_out/r5ar/tests.log
bash scripts/gpu_r5ad.sh 2>&1 | tail -11
for B in 8 32; do timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet"; done
