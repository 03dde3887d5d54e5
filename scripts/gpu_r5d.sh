#!/bin/bash
# round 5, GPU call D
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5d
O=gpurun_out/r5d
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_captured_step_gpu.py tests/test_w8pt_gpu.py -x -q > $O/gputest_sel.log 2>&1; echo "pytest rc $?" >> $O/gputest_sel.log
timeout 300 python scripts/ab_fit_sizes.py ab_libs/libdfepe_r4.so pytorch-deepfepe_amd/libdfepe_hip.so 4096 8192 16384 32768 > $O/ab_fit_default.log 2>&1
timeout 300 python bench.py --config 4 --scaling strong --no-extras --no-cpu-baseline --steps 100 > $O/bench_c4_strong.json 2> $O/bench_c4_strong.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-model > $O/bench_c3.json 2> $O/bench_c3.err
tail -4 $O/gputest_sel.log; cat $O/ab_fit_default.log
python -c "
import json
for f in ('bench_c4_strong','bench_c3'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d.get('layers_batched'))
    if d.get('api_path'):
        for k,v in d['api_path'].items():
            if isinstance(v,dict): print(k, {kk:v[kk] for kk in v if 'ms_per' in kk})
"
