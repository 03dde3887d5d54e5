#!/bin/bash
# round 5, call AS: the whole GPU suite and smoke on the final tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5as
O=gpurun_out/r5as
timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
grep -E "passed|failed" $O/gputest.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
