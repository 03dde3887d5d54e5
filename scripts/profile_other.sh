#!/bin/bash
# rocprofv3 passes for the kernels outside the bench line (config 5, the estimator): kernel trace + FETCH/WRITE/SQ counters in
# separate passes.   usage (GPU box): bash scripts/profile_other.sh r03 ; then here: python scripts/profile_other_summary.py r03
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_other_$TAG; rm -rf $O; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python $R/scripts/pmc_probe_other.py > $O/trace.log 2>&1
pmc() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python $R/scripts/pmc_probe_other.py > $O/pmc_$name.log 2>&1; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pmc mfma SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
find $O -name "*.csv" | wc -l
