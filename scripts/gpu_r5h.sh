#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5h
O=gpurun_out/r5h
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
tail -6 $O/gputest.log
python -c "
import json
d=json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], d.get('layers_batched'), d.get('full_model'))
for k,v in d['api_path'].items():
    if isinstance(v,dict): print(k, {kk:v[kk] for kk in v if 'ms_per' in kk})
"
