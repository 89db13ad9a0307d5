#!/bin/bash
out=gpurun_out/cfg3_nch.txt
: > $out
for skew in "" "--skew"; do
for nch in default 12 16 24 32; do
  echo "== cfg3 $skew MF_SWEEP_NCH=$nch" >> $out
  if [ $nch = default ]; then unset MF_SWEEP_NCH; else export MF_SWEEP_NCH=$nch; fi
  MF_ITER_MODE=sweeps python bench.py --config cfg3 $skew --steps 200 --warmup 20 --no-cpu-baseline --no-recommend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.4f frac %.3f item %.4f user %.4f %s' % (d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], r['kernel'][30:120]))" >> $out
done
done
cat $out
