#!/bin/bash
# rocprofv3 kernel trace of the other BASELINE configs (5: fit + cheirality at N = 1000, B = 4096 and 512; 2: one fit at B = 1024)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_cfg; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o b -- python $R/bench.py --no-extras --cpu-sample 8 --steps 200 --warmup 20 "$@" > $O/$name.log 2>&1; grep '^{"metric"' $O/$name.log > $O/$name.json; }
run c5 --config 5
run c5_512 --config 5 --batch 512
run c2 --config 2
python - <<PY
import csv, json
for n in ("c5", "c5_512", "c2"):
    try:
        d = json.loads(open("$O/%s.json" % n).read())
        print(n, d["value"], d["unit"], d["ms_per_step"], d["config"]["workload"][:80])
        rows = list(csv.DictReader(open("$O/%s/b_kernel_stats.csv" % n)))
        for r in rows[:5]: print("   ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
    except Exception as e:
        print(n, "failed", e); print(open("$O/%s.log" % n).read()[-500:])
PY
