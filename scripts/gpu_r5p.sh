#!/bin/bash
# round 5, call P: estimator tests on the fp16 forward (three workgroups per CU), per-kernel trace of one call, full-model bench field
mkdir -p gpurun_out/r5p
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r5p/est_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_est -- python /root/repo/scripts/est_profile.py 4096 > /dev/null 2>&1
F=$(find /tmp/prof_est -name "*kernel_stats.csv" | head -1)
cp "$F" /root/repo/gpurun_out/r5p/est_kernel_stats.csv
cd /root/repo
timeout 600 python scripts/full_model_time.py > gpurun_out/r5p/full_model.log 2>&1
tail -n 12 gpurun_out/r5p/est_tests.log gpurun_out/r5p/full_model.log; head -30 gpurun_out/r5p/est_kernel_stats.csv | cut -c1-150
