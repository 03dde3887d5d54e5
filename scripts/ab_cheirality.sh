#!/bin/bash
# Bounds of the adaptive cheirality kernel: builds of the library with every group of 64 correspondences forced through the fp32-only
# path (-DDFEPE_CHEIR_ALWAYS_FAST: lower bound) and through the fp64 stage (-DDFEPE_CHEIR_ALWAYS_SLOW: upper bound) next to the
# product build, timed on BASELINE config 5.   build here:  bash scripts/ab_cheirality.sh build ; on the GPU box: bash scripts/ab_cheirality.sh
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for v in FAST SLOW; do
    rm -rf /tmp/ab_$v && mkdir -p /tmp/ab_$v && cp -r pytorch-deepfepe_amd include /tmp/ab_$v/ && rm -rf /tmp/ab_$v/pytorch-deepfepe_amd/csrc/build
    (cd /tmp/ab_$v && DFEPE_EXTRA_FLAGS="-DDFEPE_CHEIR_ALWAYS_$v" python pytorch-deepfepe_amd/build.py > /dev/null) && cp /tmp/ab_$v/pytorch-deepfepe_amd/libdfepe_hip.so ab_libs/libdfepe_cheir_$v.so
  done
  ls -la ab_libs/*.so; exit 0
fi
for b in 4096 512; do
  for L in pytorch-deepfepe_amd/libdfepe_hip.so ab_libs/libdfepe_cheir_FAST.so ab_libs/libdfepe_cheir_SLOW.so; do
    DFEPE_LIB_PATH=$(realpath $L) timeout 200 python bench.py --config 5 --batch $b --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$b', '$(basename $L)', 'ms', d['ms_per_step'], 'fit_us', d['roofline']['avg_kernel_us'], d['accuracy'].get('pairs_with_a_valid_pose'))"
  done
done
