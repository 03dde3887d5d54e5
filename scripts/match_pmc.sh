#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mp; rm -rf $O; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $O/mfma_counters.txt
cat > /tmp/mp.py <<PY
import importlib, os, sys, torch
sys.path.insert(0, "$R")
d = importlib.import_module("pytorch-deepfepe_amd")
g = torch.Generator().manual_seed(0)
a = torch.nn.functional.normalize(torch.randn(64, 1024, 256, generator=g), dim=2).cuda()
b = torch.nn.functional.normalize(torch.randn(64, 1024, 256, generator=g), dim=2).cuda()
for _ in range(3): d.ops.nn_match_two_way(a, b, 0.7)
torch.cuda.synchronize()
PY
run() { name=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o p -- python /tmp/mp.py > $O/$name.log 2>&1; }
run a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES
run b SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES
python - <<PY
import csv, collections, glob
for sub in ("a","b"):
    try:
        rows=list(csv.DictReader(open("$O/%s/p_counter_collection.csv"%sub)))
    except Exception as e:
        print(sub, "failed", e); print(open("$O/%s.log"%sub).read()[-600:]); continue
    by=collections.OrderedDict()
    for r in rows:
        if "nn_match_tile" not in r["Kernel_Name"]: continue
        by.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
        by[r["Dispatch_Id"]]["dur_us"]=(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3
    for k,v in list(by.items())[-1:]:
        print(sub, v)
print(open("$O/mfma_counters.txt").read())
PY
