#!/bin/bash
# cfg3 power-law: rows beside the extreme ones through the cooperative kernel (MF_SWEEP_REST=coop), chunk sizes
out=gpurun_out/skew_knobs.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" MF_ITER_MODE=sweeps python bench.py --config cfg3 --skew --steps 200 --warmup 20 --no-cpu-baseline --no-recommend --check 2>>$out | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.4f frac %.3f item %.4f user %.4f check %s' % (d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], d.get('check')))" >> $out
}
run A=0
run MF_SWEEP_REST=coop
run MF_SWEEP_REST=coop MF_SWEEP_NCH=2
run MF_SWEEP_REST=coop MF_SWEEP_NCH=6
run MF_SWEEP_REST=coop MF_SWEEP_NCH=13
run MF_SWEEP_REST=coop MF_SWEEP_LONG=1500
run MF_SWEEP_REST=coop MF_SWEEP_LONG=3000
cat $out
