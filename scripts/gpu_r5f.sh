#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5f
for e in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1"; do
  echo "== env: $e"; env $e DBG_B=512 timeout 100 python scripts/est_capture_debug.py 2>&1 | grep "replay [012] " | cut -c1-200
done | tee gpurun_out/r5f/est_env.log
