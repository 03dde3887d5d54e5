cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt2; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python $R/bench.py --no-extras --cpu-sample 8 --steps 200 > $O/log.txt 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/k_kernel_stats.csv")))
for r in rows[:8]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
