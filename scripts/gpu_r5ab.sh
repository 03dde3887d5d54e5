#!/bin/bash
# round 5, call AB: same-box A/B of the gamma == 0 fix merged at the end of the backward vs layer by layer
mkdir -p gpurun_out/r5ab
cd /root/repo
for rep in 1 2; do
  for B in 8 32; do
    timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" | sed 's/^/merged:    /' >> gpurun_out/r5ab/small.log
    DFEPE_EST_FIX_AT_END_BYTES=0 timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" | sed 's/^/per layer: /' >> gpurun_out/r5ab/small.log
  done
done
cat gpurun_out/r5ab/small.log
