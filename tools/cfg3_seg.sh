#!/bin/bash
out=gpurun_out/cfg3_seg.txt
: > $out
run() { MF_ITER_MODE=sweeps python bench.py "$@" --no-cpu-baseline --no-recommend 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.4f frac %.3f item %.4f user %.4f %s' % (d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], r['kernel'][60:130]))"; }
for few in 4096 2048; do
  echo "== cfg3 uniform MF_SWEEP_FEW=$few" >> $out; MF_SWEEP_FEW=$few run --config cfg3 --steps 200 --warmup 20 >> $out
done
for seg in 256 128 64 32; do
  for long in default 400; do
    echo "== cfg3 --skew MF_SWEEP_SEG=$seg MF_SWEEP_LONG=$long" >> $out
    if [ $long = default ]; then unset MF_SWEEP_LONG; else export MF_SWEEP_LONG=$long; fi
    MF_SWEEP_FEW=2048 MF_SWEEP_SEG=$seg run --config cfg3 --skew --steps 200 --warmup 20 >> $out
  done
done
unset MF_SWEEP_LONG
for seg in 256 64; do echo "== nflx MF_SWEEP_SEG=$seg" >> $out; MF_SWEEP_SEG=$seg run --config nflx --steps 5 --warmup 2 >> $out; done
cat $out
