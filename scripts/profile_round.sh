#!/bin/bash
# All rocprofv3 passes behind profiles/rNN_*: run on the GPU box (gpurun), then scripts/profile_summary.py here.
#   usage: bash scripts/profile_round.sh r01
# PMC counters are collected in their own passes with --kernel-trace only (see MI355X_MICROARCH.md, HBM section).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 200 --warmup 20 --cpu-sample 128 > $O/bench_trace.log 2>&1
grep '^{"metric"' $O/bench_trace.log > $O/bench_line.json
pmc() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python $R/scripts/pmc_probe.py > $O/pmc_$name.log 2>&1; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pmc clk GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES
find $O -name "*.csv" | wc -l; cut -c1-200 $O/bench_line.json
