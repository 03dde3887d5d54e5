#!/bin/bash
mkdir -p gpurun_out/r5ah; cd /root/repo
timeout 300 python scripts/small_batch_profile.py 8 > gpurun_out/r5ah/profile.log 2>&1
head -75 gpurun_out/r5ah/profile.log | cut -c1-170
