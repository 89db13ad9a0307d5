#!/bin/bash
# cfg4's shape at small K: the 128-byte row pitch against rows packed at 8K bytes (MF_ROW_PITCH=0)
out=gpurun_out/smallk.txt
: > $out
for k in 10 20 30 50; do
  for pitch in 1 0; do
    echo "== K=$k MF_ROW_PITCH=$pitch" >> $out
    MF_ROW_PITCH=$pitch python bench.py --config cfg4 --feats $k --steps 10 --warmup 2 --no-cpu-baseline --no-recommend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.3f  frac %.3f  item %.3f ms  user %.3f ms  %s' % (d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], r['kernel'][:70]))" >> $out
  done
done
cat $out
