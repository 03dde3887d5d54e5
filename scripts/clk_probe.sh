#!/bin/bash
# shader clock actually sustained by each hot-path kernel: busy cycles (PMC) / duration (kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/clk; rm -rf $O; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O -o p -- python $R/scripts/pmc_probe.py > $O/log.txt 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$O/p_counter_collection.csv")))
by=collections.OrderedDict()
for r in rows:
    k=(r["Dispatch_Id"], r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:40])
    by.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
    by[k]["dur_us"]=(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3
seen={}
for (d,k),v in by.items(): seen[k]=v
for k,v in seen.items():
    if "dur_us" in v and v.get("GRBM_GUI_ACTIVE"):
        print(f'{k:42s} dur {v["dur_us"]:8.2f} us  GUI_ACTIVE {v["GRBM_GUI_ACTIVE"]:.0f}  -> {v["GRBM_GUI_ACTIVE"]/v["dur_us"]/1e3:.2f} GHz(if 1 instance)  BUSY_CU/256 {v.get("SQ_BUSY_CU_CYCLES",0)/256:.0f} wave_cycles/wave {v.get("SQ_WAVE_CYCLES",0)/max(v.get("SQ_WAVES",1),1):.0f}')
PY
