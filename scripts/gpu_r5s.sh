#!/bin/bash
# round 5, call S: prefetch depth of the fused adjoint (2 / 4 / 5 tiles in flight) and the unfused backward, same box
mkdir -p gpurun_out/r5s
cd /root/repo
for rep in 1 2; do
  DFEPE_EST_FUSE_DGRAD=0 timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5s/ab.log
  DFEPE_LIB_PATH=/root/repo/ab_libs/libdfepe_d2.so timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5s/ab.log
  timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5s/ab.log
  DFEPE_LIB_PATH=/root/repo/ab_libs/libdfepe_d5.so timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5s/ab.log
done
cat gpurun_out/r5s/ab.log
