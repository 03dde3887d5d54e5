#!/bin/bash
# round 5, call Y: kernel trace of the full model's step at B = 8 after the launch trimming
mkdir -p gpurun_out/r5y
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -- python /root/repo/scripts/small_batch_time.py 8 > /dev/null 2>&1
F=$(find /tmp/prof_small -name "*kernel_stats.csv" | head -1); cp "$F" /root/repo/gpurun_out/r5y/kernel_stats_B8.csv
cd /root/repo
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5y/kernel_stats_B8.csv')))
calls=sum(int(r['Calls']) for r in rows); tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel calls', calls, 'total kernel time per step (48 steps) us', tot/48e3)
for r in rows[:32]: print(r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us  {int(r['TotalDurationNs'])/48e3:7.1f} us/step", r['Name'][:90])
PY
