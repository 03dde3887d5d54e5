#!/bin/bash
# pipe utilisation of w8pt_fwd: busy cycles of the VALU / LDS / scalar units against the CU-busy cycles (two PMC passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/util; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o p -- python $R/scripts/inst_probe.py > $O/$name.log 2>&1; }
run a SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES
run b SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CU_CYCLES
run c SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32
python - <<PY
import csv, collections
for sub in ("a","b","c"):
    try: rows=list(csv.DictReader(open("$O/%s/p_counter_collection.csv"%sub)))
    except Exception as e:
        print(sub,"failed"); print(open("$O/%s.log"%sub).read()[-400:]); continue
    by=collections.OrderedDict()
    for r in rows:
        if "w8pt_fwd" not in r["Kernel_Name"]: continue
        by.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
        by[r["Dispatch_Id"]]["dur_us"]=(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3
    first=list(by.values())[0]
    print(sub, {k:(round(v,1) if k=="dur_us" else int(v)) for k,v in first.items()})
PY
