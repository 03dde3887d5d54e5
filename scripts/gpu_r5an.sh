#!/bin/bash
# round 5, call AN: the bench line of the final tree (driver flags, then default)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5an
O=gpurun_out/r5an
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json
for f in ('bench_driver','bench_default'):
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], (d.get('full_model') or {}).get('ms_per_step'), (d.get('full_model') or {}).get('small_batch'), {k:v for k,v in d['api_path']['pose_gt_in_loss_params'].items() if 'ms' in k})
    except Exception as e: print(f, 'ERR', e)
"
