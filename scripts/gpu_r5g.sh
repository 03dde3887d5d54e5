#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5g
O=gpurun_out/r5g
# late setting: after `import torch`, before the first CUDA call
timeout 100 python - > $O/late_env.log 2>&1 <<'PY'
import os, torch
os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
os.environ["DBG_B"] = "512"
import runpy, sys
sys.argv = ["x"]
runpy.run_path("scripts/est_capture_debug.py", run_name="__main__")
PY
grep "replay [012] " $O/late_env.log | cut -c1-160
for i in 1 2; do
timeout 300 python bench.py --steps 300 --no-extras --no-cpu-baseline > $O/bench_plain_$i.json 2>/dev/null
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python bench.py --steps 300 --no-extras --no-cpu-baseline > $O/bench_nopkt_$i.json 2>/dev/null
done
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 python -m pytest tests/test_captured_step_gpu.py -x -q > $O/gputest_captured.log 2>&1; echo "pytest rc $?" >> $O/gputest_captured.log
python -c "
import json
for f in ('bench_plain_1','bench_nopkt_1','bench_plain_2','bench_nopkt_2'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['block_stats']['median_ms_per_step'])
"
tail -5 $O/gputest_captured.log
