#!/bin/bash
# per-kernel average durations of the fused hot-path step (rocprofv3 kernel trace), printed as a table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python $R/scripts/pmc_probe.py ${1:-4096} ${2:-100} > $O/log.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/k_kernel_stats.csv")))
for r in rows[:12]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
