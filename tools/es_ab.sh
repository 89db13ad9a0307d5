#!/bin/bash
# instML100k: the two iteration forms, and the slice widths of the resident streams
out=gpurun_out/es_ab.txt
: > $out
run() { python bench.py --config ml100k --steps 3000 --warmup 200 --no-cpu-baseline --no-recommend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.4f  value %.3e  frac %.3f  item|errors %.4f ms  user|streams %.4f ms  %s' % (d['ms_per_step'], d['value'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], r['kernel'][r['kernel'].index('iterate='):]))"; }
echo "== sweeps" >> $out; MF_ITER_MODE=sweeps run >> $out
for sw in 8 4; do echo "== errors + resident streams, MF_ES_SW=$sw" >> $out; MF_ES_SW=$sw run >> $out; done
cat $out
