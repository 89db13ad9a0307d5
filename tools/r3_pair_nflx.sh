#!/bin/bash
# Netflix shape / cfg4-Zipf: the wave-pair form on the ITEM side (long rows everywhere) against the single-wave form, over the
# extreme-row threshold (experiments build)
R=${GRAFT_REPO_ROOT:-.}; cd $R; export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
run() { env "$@" python3 bench.py $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-recommend --check 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; c=d.get('check') or {}
print('%-60s ms %8.3f frac %.3f item %7.3f user %7.3f  %s %s' % ('$*', d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], c.get('L_bit_identical'), r['kernel'].split('long_rows=')[1][:24]))"; }
CFG="--config nflx"
run MF_X=0
for t in 40000 80000 160000 400000; do run MF_SWEEP_PAIR_I=1 MF_SWEEP_LONG_I=$t; done
run MF_SWEEP_PAIR_I=1
CFG="--columns zipf"
run MF_X=0
for t in 80000 160000 400000 1000001; do run MF_SWEEP_PAIR_I=1 MF_SWEEP_LONG_I=$t; done
