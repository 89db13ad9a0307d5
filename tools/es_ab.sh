#!/bin/bash
# A/B of the two iteration forms (errors + streams vs the two sweeps) on the cache-resident BASELINE configs.
# usage: gpurun -- bash tools/es_ab.sh   -> gpurun_out/es_ab.txt
out=gpurun_out/es_ab.txt
: > $out
for cfg in "ml100k --steps 3000 --warmup 200"; do
  for mode in es sweeps; do
    echo "== $cfg MF_ITER_MODE=$mode" >> $out
    MF_ITER_MODE=$mode python bench.py --config $cfg --no-cpu-baseline --no-recommend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.4f  value %.3e  frac %.3f  item/errors %.4f ms  user/streams %.4f ms  %s' % (d['ms_per_step'], d['value'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], r['kernel'][-90:]))" >> $out
  done
done
cat $out
