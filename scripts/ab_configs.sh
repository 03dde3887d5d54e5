for r in 1 2; do
for L in ab_libs/prod.so ab_libs/lean16.so; do
  for cfg in "--config 5" "--config 5 --batch 512" "--config 4 --scaling strong"; do
    DFEPE_LIB_PATH=$(realpath $L) timeout 300 python bench.py $cfg --no-extras --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$(basename $L)', '$cfg', 'ms', d['ms_per_step'], 'blocks', d['block_stats']['median_ms_per_step'])"
  done
done
done
