#!/bin/bash
# round 5, call AD: per-kernel times of one estimator call at B = 8, GEMMs as built for full grids vs the small-grid build
mkdir -p gpurun_out/r5ad
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  rm -rf /tmp/prof_e
  DFEPE_EST_SMALL_GRID=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -- python /root/repo/scripts/est_profile.py 8 > /dev/null 2>&1
  F=$(find /tmp/prof_e -name "*kernel_stats.csv" | head -1); cp "$F" /root/repo/gpurun_out/r5ad/stats_mode$mode.csv
done
cd /root/repo
python - <<'PY'
import csv
for mode in (0,1):
    rows=list(csv.DictReader(open(f'gpurun_out/r5ad/stats_mode{mode}.csv')))
    print('mode', mode)
    for r in rows:
        if 'est_gemm' in r['Name']: print('  ', r['Calls'].rjust(4), f"avg {float(r['AverageNs'])/1e3:7.1f} min {int(r['MinNs'])/1e3:7.1f} max {int(r['MaxNs'])/1e3:7.1f}", r['Name'][28:75])
PY
