#!/bin/bash
# round 5, call AO: last check of the final tree: the whole GPU suite, smoke, the kernels outside the bench line under rocprofv3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5ao
O=gpurun_out/r5ao
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
grep -E "passed|failed" $O/gputest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(cd /tmp && GRAFT_REPO_ROOT=$R bash $R/scripts/profile_other.sh r05 2>&1 | tail -1)
