#!/bin/bash
# round 5, call V: the whole GPU suite, smoke, the bench line (default and driver flags) and the kernels outside the bench line
# (config 5, the estimator) under rocprofv3 -- after the estimator's fp16 forward and fused data gradient
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5v
O=gpurun_out/r5v
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
tail -4 $O/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
(cd /tmp && GRAFT_REPO_ROOT=$R bash $R/scripts/profile_other.sh r05 2>&1 | tail -2)
python -c "
import json
for f in ('bench_driver','bench_default'):
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], (d.get('full_model') or {}))
    except Exception as e: print(f, 'ERR', e)
"
