#!/bin/bash
# round 5, call U: the K loop with the LDS stage filled in two halves (each under the other half's MFMAs) against issue / wait / multiply
mkdir -p gpurun_out/r5u
cd /root/repo
timeout 600 python -m pytest tests/test_estimator_mfma_gpu.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r5u/est_tests.log
for rep in 1 2; do
  DFEPE_LIB_PATH=/root/repo/ab_libs/libdfepe_nosplit.so timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5u/ab.log
  timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5u/ab.log
done
cat gpurun_out/r5u/est_tests.log gpurun_out/r5u/ab.log
