#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5m
O=gpurun_out/r5m
export TMPDIR=/tmp
R=$PWD
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for c in 2 4 5; do timeout 200 python bench.py --config $c --no-extras > $O/bench_c$c.json 2>/dev/null; done
timeout 200 python bench.py --config 5 --batch 512 --no-extras > $O/bench_c5_512.json 2>/dev/null
timeout 200 python bench.py --config 4 --scaling strong --no-extras --no-cpu-baseline > $O/bench_c4_strong.json 2>/dev/null
timeout 200 python bench.py --force-dist --no-extras --no-cpu-baseline > $O/bench_force_dist.json 2>/dev/null
timeout 400 python scripts/batch_sweep.py $R/gpurun_out/r5m/r05_batch_sweep.md > $O/batch_sweep.log 2>&1
(cd /tmp && GRAFT_REPO_ROOT=$R bash $R/scripts/profile_other.sh r05 2>&1 | tail -2)
python -c "
import json
for f in ('bench_driver','bench_default','bench_c2','bench_c4','bench_c5','bench_c5_512','bench_c4_strong','bench_force_dist'):
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['roofline']['avg_kernel_us'], (d.get('full_model') or {}).get('ms_per_step'), (d.get('full_model') or {}).get('small_batch'))
    except Exception as e: print(f, 'ERR', e)
"
tail -3 $O/batch_sweep.log
