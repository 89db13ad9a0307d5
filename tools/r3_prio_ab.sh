#!/bin/bash
# round 3: lean issue + wave priority for long rows: thresholds of the extreme-row path re-swept
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
run() { name=$1; shift; env "$@" python3 bench.py --config ${CFG:-cfg3} $SKEW --steps ${STEPS:-200} --warmup ${WARM:-20} --no-cpu-baseline --no-recommend $CHECK > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; }
  python3 - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    k = r["kernel"]; c = d.get("check") or {}
    print("%-30s ms %8.4f frac %.3f item %.4f user %.4f bit-identical %s %s | long%s mid%s" % (sys.argv[2], d["ms_per_step"], r["frac"], r.get("item_sweep_ms") or 0, r.get("user_sweep_ms") or 0,
          c.get("L_bit_identical"), c.get("R_bit_identical"), k.split("long_rows")[1].split()[0], k.split("mid_rows")[1].split()[0]))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
CHECK=--check
SKEW=--skew
run pl_rule
run pl_prio0 MF_SWEEP_PRIO=0
for q in 128 256 400; do run pl_prio$q MF_SWEEP_PRIO=$q; done
for t in 1200 1700 2400 3500 6000; do run pl_long${t} MF_SWEEP_LONG=$t; run pl_long${t}_prio0 MF_SWEEP_LONG=$t MF_SWEEP_PRIO=0; done
run pl_noskew MF_SWEEP_SKEW=0
run pl_noskew_prio0 MF_SWEEP_SKEW=0 MF_SWEEP_PRIO=0
SKEW=
run un_rule
run un_prio200 MF_SWEEP_PRIO=200
CFG=cfg4 STEPS=10 WARM=2 CHECK=
run cfg4_a
run cfg4_pf0 MF_SWEEP_PF=0
run cfg4_b
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
