#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_w8pt_gpu.py -q -k lean 2>&1 | grep -n "^E  \|passed\|failed" | cut -c1-200 | head
