#!/bin/bash
mkdir -p gpurun_out/r5am; cd /root/repo
for rep in 1 2; do
  timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" | sed 's/^/pass:       /' >> gpurun_out/r5am/ab.log
  DFEPE_EST_PASS=0 timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" | sed 's/^/per launch: /' >> gpurun_out/r5am/ab.log
done
cat gpurun_out/r5am/ab.log
