#!/bin/bash
# round 5, call AC: the GEMMs' small-grid build (B fragments two tiles ahead) at the reference's batch sizes, same box A/B
mkdir -p gpurun_out/r5ac
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r5ac/tests.log
for rep in 1 2; do
  for B in 8 32; do
    timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" | sed 's/^/ahead 2:  /' >> gpurun_out/r5ac/small.log
    DFEPE_EST_SMALL_GRID=0 timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" | sed 's/^/as large: /' >> gpurun_out/r5ac/small.log
  done
done
cat gpurun_out/r5ac/tests.log gpurun_out/r5ac/small.log
