#!/bin/bash
# round 5, call X: table-driven weight preparation + one reduction launch per backward: tests, small-batch and large-batch timing
mkdir -p gpurun_out/r5x
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py tests/test_captured_step_gpu.py tests/test_compat_gpu.py -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r5x/tests.log
for B in 8 32; do timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" >> gpurun_out/r5x/small.log; done
timeout 200 python scripts/small_batch_time.py 8 1000 2>&1 | grep "full DeepFNet" >> gpurun_out/r5x/small.log
timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5x/small.log
cat gpurun_out/r5x/tests.log gpurun_out/r5x/small.log
