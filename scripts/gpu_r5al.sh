#!/bin/bash
# round 5, call AL: final check after the one-call-per-pass estimator: GPU suite, smoke, stress sweep, bench lines, small-batch table
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5al
O=gpurun_out/r5al
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
grep -E "passed|failed" $O/gputest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python scripts/stress_estimator.py 60 2 > $O/stress.log 2>&1; tail -1 $O/stress.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for B in 8 32; do timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" >> $O/small.log; done
timeout 200 python scripts/small_batch_time.py 8 1000 2>&1 | grep "full DeepFNet" >> $O/small.log
timeout 200 python scripts/estimator_time.py 4096 100 2>&1 | grep "stock\|fused\|split" >> $O/estimator_time.log
timeout 200 python scripts/estimator_time.py 12 2000 2>&1 | grep "stock\|split" >> $O/estimator_time.log
cat $O/small.log $O/estimator_time.log
python -c "
import json
for f in ('bench_driver','bench_default'):
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], (d.get('full_model') or {}).get('ms_per_step'), (d.get('full_model') or {}).get('small_batch'))
    except Exception as e: print(f, 'ERR', e)
"
