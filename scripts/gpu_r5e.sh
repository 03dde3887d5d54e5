#!/bin/bash
# round 5, GPU call E
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5e
O=gpurun_out/r5e
export TMPDIR=/tmp
timeout 300 python scripts/capture_probe2.py > $O/capture_probe2.log 2>&1
timeout 300 python -m pytest tests/test_captured_step_gpu.py -x -q > $O/gputest_captured.log 2>&1; echo "pytest rc $?" >> $O/gputest_captured.log
timeout 200 python scripts/ab_fit_sizes.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 4096 8192 > $O/ab_fit.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_captured_step_gpu.py > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
cat $O/capture_probe2.log; tail -4 $O/gputest_captured.log; cat $O/ab_fit.log; tail -4 $O/gputest.log
