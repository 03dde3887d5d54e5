#!/bin/bash
# round 5, call Z: faster weight preparation / reduction kernels; split-stage K loop at the reference's batch sizes
mkdir -p gpurun_out/r5z
cd /root/repo
timeout 900 python -m pytest tests/test_estimator_mfma_gpu.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r5z/tests.log
for B in 8 32; do
  timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" >> gpurun_out/r5z/small.log
  DFEPE_LIB_PATH=/root/repo/ab_libs/libdfepe_split.so timeout 200 python scripts/small_batch_time.py $B 2>&1 | grep "full DeepFNet" | sed 's/^/split: /' >> gpurun_out/r5z/small.log
done
timeout 200 python scripts/est_ab.py 2>&1 | grep "lib=" >> gpurun_out/r5z/small.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -- python /root/repo/scripts/small_batch_time.py 8 > /dev/null 2>&1
F=$(find /tmp/prof_small -name "*kernel_stats.csv" | head -1); cp "$F" /root/repo/gpurun_out/r5z/kernel_stats_B8.csv
cd /root/repo
cat gpurun_out/r5z/tests.log gpurun_out/r5z/small.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5z/kernel_stats_B8.csv')))
calls=sum(int(r['Calls']) for r in rows); tot=sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel calls', calls, 'total kernel time per step (48 steps) us', tot/48e3)
for r in rows[:14]: print(r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.1f} us  {int(r['TotalDurationNs'])/48e3:7.1f} us/step", r['Name'][:90])
PY
