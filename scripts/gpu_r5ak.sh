#!/bin/bash
mkdir -p gpurun_out/r5ak; cd /root/repo
timeout 600 python -m pytest tests/test_estimator_mfma_gpu.py tests/test_estimator_gpu.py tests/test_captured_step_gpu.py tests/test_compat_gpu.py -q -m gpu 2>&1 | grep -E "^E|passed|failed|Error" | head -30 > gpurun_out/r5ak/t.log
cat gpurun_out/r5ak/t.log
for B in 8 32; do timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet"; done
