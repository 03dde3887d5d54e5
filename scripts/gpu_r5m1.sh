#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5m
O=gpurun_out/r5m
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -5 $O/gputest.log
bash scripts/profile_round.sh r05 2>&1 | tail -3
