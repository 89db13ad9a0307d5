#!/bin/bash
# round 3: A/B of the double-buffered sweep on the cfg3 shapes (experiments build), every variant with --check
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R
export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
run() { name=$1; shift; env "$@" python3 bench.py --config cfg3 $SKEW --steps 200 --warmup 20 --no-cpu-baseline --no-recommend --check > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; }
  python3 - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    print("%-28s ms %8.4f frac %.3f item %.4f user %.4f bit-identical %s %s | %s" % (sys.argv[2], d["ms_per_step"], r["frac"], r.get("item_sweep_ms") or 0, r.get("user_sweep_ms") or 0,
          d["check"]["L_bit_identical"], d["check"]["R_bit_identical"], r["kernel"].split("long_rows")[1][:60]))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
SKEW=--skew
run pl_db0 MF_SWEEP_DB=0
run pl_db16 MF_SWEEP_DB=1
run pl_db24 MF_SWEEP_DB=1 MF_SWEEP_DB_NCH=24
run pl_db32 MF_SWEEP_DB=1 MF_SWEEP_DB_NCH=32
run pl_db12 MF_SWEEP_DB=1 MF_SWEEP_DB_NCH=12
for t in 300 400 600 1200 2400; do run pl_db16_long$t MF_SWEEP_DB=1 MF_SWEEP_LONG=$t; done
for t in 400 1200; do run pl_db32_long$t MF_SWEEP_DB=1 MF_SWEEP_DB_NCH=32 MF_SWEEP_LONG=$t; done
run pl_db16_noskew MF_SWEEP_DB=1 MF_SWEEP_SKEW=0
run pl_db32_noskew MF_SWEEP_DB=1 MF_SWEEP_DB_NCH=32 MF_SWEEP_SKEW=0
SKEW=
run un_db0 MF_SWEEP_DB=0
run un_db16 MF_SWEEP_DB=1
run un_db32 MF_SWEEP_DB=1 MF_SWEEP_DB_NCH=32
