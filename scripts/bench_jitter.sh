#!/bin/bash
# How stable is the contract number at the driver's K = 20?  The timed region is 2 ms: bench.py --steps 20 --warmup 5, six times with
# and without the periodic synchronisation of the untimed spin-up steps, timed ms/step against the median of the ten informational
# blocks behind it.
for i in 1 2 3 4 5 6; do for sp in 0 1; do DFEPE_BENCH_SPINUP_SYNC=$sp python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('spinup_sync=$sp', d['ms_per_step'], d['block_stats']['median_ms_per_step'])"; done; done
