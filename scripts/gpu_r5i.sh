#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5i
O=gpurun_out/r5i
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_compat_gpu.py -q -k "legacy" > $O/legacy.log 2>&1
grep -n "^E  \|passed\|failed" $O/legacy.log | head -20
