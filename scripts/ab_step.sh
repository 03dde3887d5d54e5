#!/bin/bash
# Same-box A/B of two builds of libdfepe_hip.so on the whole bench step (box-to-box variance is ~3 %, more than most
# kernel changes): alternates `bench.py --no-extras` runs with DFEPE_LIB_PATH pointing at each library.
#   usage (GPU box): bash scripts/ab_step.sh pytorch-deepfepe_amd/libdfepe_hip_A.so pytorch-deepfepe_amd/libdfepe_hip.so [rounds]
A=$(realpath $1); B=$(realpath $2); R=${3:-3}
for r in $(seq $R); do
  for L in $A $B; do
    DFEPE_LIB_PATH=$L timeout 300 python bench.py --no-extras --cpu-sample 8 --steps 300 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); rb=d.get('recurrent_backward') or {}
print('$(basename $L)', 'step_ms', d['ms_per_step'], 'median', d['block_stats']['median_ms_per_step'], 'fit_fwd_us', d['roofline']['avg_kernel_us'], 'bwd', rb.get('w8pt_bwd_us_gF_only'), rb.get('w8pt_bwd_us_gF_gResidual_gEpi'))"
  done
done
