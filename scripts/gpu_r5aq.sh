#!/bin/bash
# round 5, call AQ: kernel trace of the full model's training step at 8 pairs per batch (final tree) -> profiles/r05_small_batch_kernels.md
mkdir -p gpurun_out/r5aq
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -- python /root/repo/scripts/small_batch_time.py 8 > /root/repo/gpurun_out/r5aq/run.log 2>&1
F=$(find /tmp/prof_small -name "*kernel_stats.csv" | head -1); cp "$F" /root/repo/gpurun_out/r5aq/kernel_stats_B8.csv
grep "full DeepFNet" /root/repo/gpurun_out/r5aq/run.log
