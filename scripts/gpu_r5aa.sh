#!/bin/bash
# round 5, call AA: gamma == 0 fixes in one launch at the end of a small backward; faster maxima
mkdir -p gpurun_out/r5aa
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py tests/test_captured_step_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r5aa/tests.log
for B in 8 32; do timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" >> gpurun_out/r5aa/small.log; done
timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5aa/small.log
cat gpurun_out/r5aa/tests.log gpurun_out/r5aa/small.log
