#!/bin/bash
# second pass: the pair form on the user side of the Netflix shape / cfg4-Zipf, on cfg4 with uniform columns, finer thresholds
R=${GRAFT_REPO_ROOT:-.}; cd $R; export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
run() { env "$@" python3 bench.py $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-recommend --check 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; c=d.get('check') or {}
print('%-72s ms %8.3f frac %.3f item %7.3f user %7.3f  %s %s' % ('$CFG $*', d['ms_per_step'], r['frac'], r['item_sweep_ms'], r['user_sweep_ms'], c.get('L_bit_identical'), r['kernel'].split('long_rows=')[1][:10]))"; }
CFG="--config nflx"
run MF_SWEEP_PAIR_I=1 MF_SWEEP_LONG_I=60000
run MF_SWEEP_PAIR_I=1 MF_SWEEP_LONG_I=100000
run MF_SWEEP_PAIR_I=1 MF_SWEEP_LONG_I=80000 MF_SWEEP_PAIR_U=1
CFG="--columns zipf"
run MF_SWEEP_PAIR_I=1 MF_SWEEP_LONG_I=120000
run MF_SWEEP_PAIR_I=1 MF_SWEEP_LONG_I=220000
run MF_SWEEP_PAIR_I=1 MF_SWEEP_LONG_I=160000 MF_SWEEP_PAIR_U=1
CFG=""
run MF_X=0
run MF_SWEEP_PAIR_I=1
run MF_SWEEP_PAIR_U=1
