#!/bin/bash
# Round-end check on the GPU box: the whole -m gpu suite, smoke(), the bench line with the driver's flags and with the defaults,
# and the one-rank distributed form.  Every step is bounded and reads nothing from stdin.   bash scripts/final_check.sh
cd "$(dirname "$0")/.."
O=gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu < /dev/null > $O/gpu_suite.log 2>&1; echo "pytest rc=$?" ; tail -3 $O/gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench(driver flags) rc=$?"
timeout 300 python bench.py < /dev/null > $O/bench_default.json 2> $O/bench_default.err; echo "bench(default) rc=$?"
timeout 200 python bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-extras --no-cpu-baseline < /dev/null > $O/bench_dist.json 2> $O/bench_dist.err; echo "bench(force-dist) rc=$?"
for f in bench_driver bench_default bench_dist; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    fm = d.get("full_model") or {}
    ap = d.get("api_path") or {}
    print(sys.argv[1], "ms", d["ms_per_step"], "value", d["value"], "frac", (d.get("roofline") or {}).get("frac"), "fit_us", (d.get("roofline") or {}).get("avg_kernel_us"),
          "full_model_ms", fm.get("ms_per_step"), "api", {k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in ap.items()} if isinstance(ap, dict) else ap,
          "exchange", (d.get("config") or {}).get("loss_exchange"), (d.get("config") or {}).get("loss_exchange_fallback"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
