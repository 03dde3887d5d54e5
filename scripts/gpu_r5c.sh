#!/bin/bash
# round 5, GPU call C
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5c
O=gpurun_out/r5c
export TMPDIR=/tmp
timeout 400 python scripts/capture_probe2.py > $O/capture_probe2.log 2>&1
timeout 300 python scripts/ab_fit_sizes.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 4096 8192 16384 32768 > $O/ab_fit_default.log 2>&1
DFEPE_FIT_LEAN=1 timeout 300 python scripts/ab_fit_sizes.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 4096 > $O/ab_fit_lean4096.log 2>&1
DFEPE_FIT_LEAN=0 timeout 300 python scripts/ab_fit_sizes.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 8192 32768 > $O/ab_fit_nolean.log 2>&1
AB_N=128 timeout 300 python scripts/ab_fit_sizes.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 4096 16384 > $O/ab_fit_n128.log 2>&1
timeout 200 python scripts/ab_config5.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 512 > $O/ab_config5_512.log 2>&1
timeout 300 python bench.py --config 4 --scaling strong --no-extras --no-cpu-baseline --steps 100 > $O/bench_c4_strong.json 2> $O/bench_c4_strong.err
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-full-model > $O/bench_c3.json 2> $O/bench_c3.err
cat $O/capture_probe2.log $O/ab_fit_default.log $O/ab_fit_lean4096.log $O/ab_fit_nolean.log $O/ab_fit_n128.log $O/ab_config5_512.log
python -c "
import json
for f in ('bench_c4_strong','bench_c3'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d.get('layers_batched'))
"
