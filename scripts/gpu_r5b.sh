#!/bin/bash
# round 5, GPU call B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
export TMPDIR=/tmp
timeout 300 python scripts/capture_probe.py > $O/capture_probe.log 2>&1
timeout 300 python scripts/ab_config5.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 4096 512 > $O/ab_config5.log 2>&1
DFEPE_POSE_LAUNCHES=2 timeout 200 python scripts/ab_config5.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 512 > $O/ab_config5_two_launches.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_captured_step_gpu.py > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
timeout 200 python scripts/eager_profile.py 300 0 1 > $O/eager_profile_ref.log 2>&1
timeout 300 python -m pytest tests/test_captured_step_gpu.py -x -q > $O/gputest_captured.log 2>&1; echo "pytest rc $?" >> $O/gputest_captured.log
cat $O/capture_probe.log; cat $O/ab_config5.log $O/ab_config5_two_launches.log; tail -3 $O/gputest.log; head -3 $O/eager_profile_ref.log; tail -5 $O/gputest_captured.log
