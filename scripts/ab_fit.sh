#!/bin/bash
# Same-box comparison of several builds of libdfepe_hip.so on the stand-alone forward fit (bench.py's roofline probe)
# and the whole step.   usage (GPU box): bash scripts/ab_fit.sh libA.so libB.so ... 
for r in 1 2; do
  for L in "$@"; do
    DFEPE_LIB_PATH=$(realpath $L) timeout 300 python bench.py --no-extras --cpu-sample 8 --steps 300 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$(basename $L)', 'step_ms', d['block_stats']['median_ms_per_step'], 'fit_fwd_us', d['roofline']['avg_kernel_us'])"
  done
done
