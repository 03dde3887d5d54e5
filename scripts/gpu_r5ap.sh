#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r5ap
timeout 400 python bench.py > gpurun_out/r5ap/bench_default.json 2> gpurun_out/r5ap/bench_default.err
python -c "
import json
d=json.load(open('gpurun_out/r5ap/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('full_model'))
"
