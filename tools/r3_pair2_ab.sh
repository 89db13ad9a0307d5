#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R
export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
run() { name=$1; shift; env "$@" python3 bench.py --config ${CFG:-cfg3} $SKEW --steps ${STEPS:-200} --warmup ${WARM:-20} --no-cpu-baseline --no-recommend $CHECK > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; }
  python3 - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    k = r["kernel"]; c = d.get("check") or {}
    print("%-30s ms %8.4f frac %.3f item %.4f user %.4f bit-identical %s %s | long%s pair%s" % (sys.argv[2], d["ms_per_step"], r["frac"], r.get("item_sweep_ms") or 0, r.get("user_sweep_ms") or 0,
          c.get("L_bit_identical"), c.get("R_bit_identical"), k.split("long_rows")[1].split()[0], k.split("wave_pair")[1].split()[0]))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
CHECK=--check
SKEW=--skew
run nl1
for n in 32 48 64; do run nl2_nch$n MF_SWEEP_PAIR_LOADERS=2 MF_SWEEP_PAIR_NCH=$n; done
run nl1_noskew MF_SWEEP_SKEW=0
for n in 32 64; do run nl2_noskew_nch$n MF_SWEEP_PAIR_LOADERS=2 MF_SWEEP_SKEW=0 MF_SWEEP_PAIR_NCH=$n; done
for t in 1200 2000; do run nl2_I${t} MF_SWEEP_PAIR_LOADERS=2 MF_SWEEP_LONG_I=$t; done
