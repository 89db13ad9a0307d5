#!/bin/bash
# timing experiments on the streams launch (results are wrong with MF_ES_DBG != 0)
out=gpurun_out/es_dbg.txt
: > $out
for dbg in 0 1 2 4 7; do
  for w in 768; do
  echo "== ml100k MF_ES_DBG=$dbg waves=$w" >> $out
  MF_ES_WAVES=$w MF_ES_DBG=$dbg MF_ITER_MODE=es python bench.py --config ml100k --steps 2000 --warmup 100 --no-cpu-baseline --no-recommend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.4f errors %.4f ms streams %.4f ms' % (d['ms_per_step'], r['item_sweep_ms'], r['user_sweep_ms']))" >> $out
  done
done
cat $out
