#!/bin/bash
# round 5, GPU call A: whole GPU suite on the new tree, config-5 A/B against the round-4 library, host profile of the eager API path,
# the bench line with the driver's flags
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5a
O=gpurun_out/r5a
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
timeout 300 python scripts/ab_config5.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 4096 512 1024 > $O/ab_config5.log 2>&1
DFEPE_POSE_LAUNCHES=2 timeout 200 python scripts/ab_config5.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 512 1024 > $O/ab_config5_two_launches.log 2>&1
timeout 200 python scripts/eager_profile.py 300 0 1 > $O/eager_profile_ref.log 2>&1
timeout 200 python scripts/eager_profile.py 300 1 1 > $O/eager_profile_gt.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
tail -3 $O/gputest.log; cat $O/ab_config5.log $O/ab_config5_two_launches.log; head -3 $O/eager_profile_ref.log
