#!/bin/bash
# cfg3 power-law and the Netflix-shaped instance after a change to the extreme-row path: parity against the plain
# sweeps (--check) and the time per iteration; then the kernel timeline of one iteration
out=gpurun_out/skew_knobs.txt
: > $out
run() {
  cfg=$1; shift
  echo "== $cfg $*" >> $out
  env "$@" MF_ITER_MODE=sweeps python bench.py $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-recommend --check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
r=d['roofline']
print('ms_per_step %.4f frac %.3f item %.4f user %.4f check %s' % (d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], d.get('check')))" >> $out
}
run "--config cfg3 --skew" A=0
run "--config cfg3 --skew" MF_SWEEP_LONG=500
run "--config cfg3 --skew" MF_SWEEP_LONG=336
run "--config nflx" A=0
cat $out
bash tools/trace_skew.sh
