#!/bin/bash
# cfg3 power-law: where to cut rows into the products + ordered-sum path (MF_SWEEP_LONG = entries)
out=gpurun_out/skew_long.txt
: > $out
for t in default 96 128 200 336 500; do
  echo "== cfg3 --skew MF_SWEEP_LONG=$t" >> $out
  if [ $t = default ]; then unset MF_SWEEP_LONG; else export MF_SWEEP_LONG=$t; fi
  MF_ITER_MODE=sweeps python bench.py --config cfg3 --skew --steps 200 --warmup 20 --no-cpu-baseline --no-recommend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.4f frac %.3f item %.4f user %.4f %s' % (d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], r['kernel'][60:140]))" >> $out
done
cat $out
