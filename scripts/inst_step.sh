#!/bin/bash
# dynamic instruction counts per wavefront of every kernel of the fused step (one PMC pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/is; rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O -o p -- python $R/scripts/pmc_probe.py 4096 100 > $O/log.txt 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$O/p_counter_collection.csv")))
by=collections.OrderedDict()
for r in rows:
    by.setdefault((r["Dispatch_Id"],r["Kernel_Name"][:70]),{})[r["Counter_Name"]]=float(r["Counter_Value"])
seen=set()
for (d,k),v in by.items():
    if k in seen: continue
    seen.add(k)
    wv=max(v.get("SQ_WAVES",1),1)
    print(k, "waves", int(wv), {n: round(x/wv,1) for n,x in v.items() if n!="SQ_WAVES"})
PY
