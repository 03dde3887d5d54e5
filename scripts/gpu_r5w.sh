#!/bin/bash
# round 5, call W: the full model's step at the reference's batch sizes, and its launch count
mkdir -p gpurun_out/r5w
cd /root/repo
for B in 8 32; do timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" >> gpurun_out/r5w/small.log; done
timeout 200 python scripts/small_batch_time.py 8 1000 2>&1 | grep "full DeepFNet" >> gpurun_out/r5w/small.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -- python /root/repo/scripts/small_batch_time.py 8 > /dev/null 2>&1
F=$(find /tmp/prof_small -name "*kernel_stats.csv" | head -1); cp "$F" /root/repo/gpurun_out/r5w/kernel_stats_B8.csv
cd /root/repo; cat gpurun_out/r5w/small.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5w/kernel_stats_B8.csv')))
calls=sum(int(r['Calls']) for r in rows); print('total kernel calls', calls, '(4+20 eager steps, 1 warm-up + capture, 23 replays...)')
for r in rows[:25]: print(r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us", r['Name'][:100])
PY
