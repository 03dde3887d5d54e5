#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5k
O=gpurun_out/r5k
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 400 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['block_stats']['median_ms_per_step'], d['roofline']['avg_kernel_us'])"; }
run base X=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run hwq1 GPU_MAX_HW_QUEUES=1
run hwq2 GPU_MAX_HW_QUEUES=2
run kacopy0 DEBUG_HIP_KERNARG_COPY_OPT=0
run base2 X=1
timeout 300 python -m pytest tests/test_w8pt_gpu.py -q -k lean 2>&1 | tail -1
