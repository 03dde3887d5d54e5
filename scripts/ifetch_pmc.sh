#!/bin/bash
# instruction-fetch behaviour of the hot kernels (straight-line code, I-cache cold at every launch?): two PMC passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ifetch; rm -rf $O; mkdir -p $O
pmc() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o p -- python $R/scripts/pmc_probe.py > $O/$name.log 2>&1; }
pmc ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ
pmc sq SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES
python - <<PY
import csv, collections
for sub in ("ic", "sq"):
    try: rows = list(csv.DictReader(open("$O/%s/p_counter_collection.csv" % sub)))
    except Exception as e:
        print(sub, "failed", e); print(open("$O/%s.log" % sub).read()[-600:]); continue
    by = collections.OrderedDict()
    for r in rows:
        k = (r["Kernel_Name"][:70], r["Dispatch_Id"])
        by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        by[k]["dur_us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    agg = collections.OrderedDict()
    for (name, _), v in by.items():
        if not any(s in name for s in ("w8pt16", "loss_tail")): continue
        a = agg.setdefault(name, collections.Counter()); a["n"] += 1
        for c, x in v.items(): a[c] += x
    for name, a in agg.items():
        print(sub, name, {c: round(x / a["n"], 1) for c, x in a.items() if c != "n"})
PY
