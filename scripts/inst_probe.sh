#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ip; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O -o p -- python $R/scripts/inst_probe.py > $O/log.txt 2>&1
tail -2 $O/log.txt
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$O/p_counter_collection.csv")))
by=collections.OrderedDict()
for r in rows:
    if "w8pt_fwd" not in r["Kernel_Name"]: continue
    by.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in by.items():
    wv=v.get("SQ_WAVES",1)
    print(k, {n: round(x/wv,1) for n,x in v.items()})
PY
