#!/bin/bash
# the three-wave form (loader / phase A / phase B, MF_SWEEP_TRIO=1, experiments build) against the wave pair on the shapes that
# take the pair form; every line with --check (bit-identical to the plain sweeps)
R=${GRAFT_REPO_ROOT:-.}; cd $R; export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
run() { env "$@" python3 bench.py $CFG --no-cpu-baseline --no-recommend --check 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; c=d.get('check') or {}
print('%-44s %-28s ms %9.4f frac %.3f item %8.4f user %8.4f  %s %s' % ('$CFG', '$*', d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], c.get('L_bit_identical'), c.get('R_bit_identical')))"; }
for CFG in "--config cfg3 --skew --steps 200 --warmup 20" "--config nflx --steps 10 --warmup 2" "--columns zipf --steps 10 --warmup 2" "--steps 10 --warmup 2"; do
  run MF_SWEEP_TRIO=0; run MF_SWEEP_TRIO=1; run MF_SWEEP_TRIO=1 MF_SWEEP_PAIR_NCH=24; run MF_SWEEP_TRIO=1 MF_SWEEP_PAIR_NCH=48
done
