#!/bin/bash
# round 5, call R: fused data gradient: kernel trace + repeated A/B timing (order swapped)
mkdir -p gpurun_out/r5r
cd /root/repo
DFEPE_EST_FUSE_DGRAD=0 timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16\|Error\|error" > gpurun_out/r5r/time_separate_1.log
timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16\|Error\|error" > gpurun_out/r5r/time_fused_1.log
DFEPE_EST_FUSE_DGRAD=0 timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16\|Error\|error" > gpurun_out/r5r/time_separate_2.log
timeout 300 python scripts/estimator_time.py 4096 100 2>&1 | grep "split-bf16\|Error\|error" > gpurun_out/r5r/time_fused_2.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_est -- python /root/repo/scripts/est_profile.py 4096 > /dev/null 2>&1
F=$(find /tmp/prof_est -name "*kernel_stats.csv" | head -1)
cp "$F" /root/repo/gpurun_out/r5r/est_kernel_stats_fused.csv
cd /root/repo
tail -n 3 gpurun_out/r5r/*.log; head -8 gpurun_out/r5r/est_kernel_stats_fused.csv | cut -c1-200
