#!/bin/bash
# round 5, call AG: the fused adjoint with x^ folded into per-(channel, pair) constants (two FMAs per element in pass B)
mkdir -p gpurun_out/r5ag
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r5ag/tests.log
for rep in 1 2 3; do timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5ag/ab.log; done
cat gpurun_out/r5ag/tests.log gpurun_out/r5ag/ab.log
