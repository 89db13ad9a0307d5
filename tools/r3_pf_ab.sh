#!/bin/bash
# round 3: A/B of the pipelined phases (MF_SWEEP_PF) on cfg4 / cfg3 uniform / cfg3 power-law, and of the mid-row launch with them
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3d; mkdir -p $O; cd $R
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
run pl_pf0 MF_SWEEP_MID=0
run pl_pf1 MF_SWEEP_MID=0 MF_SWEEP_PF=1
run pl_pf1_nch32 MF_SWEEP_MID=0 MF_SWEEP_PF=1 MF_SWEEP_NCH=32
for m in 192 384; do for n in 16 32 48; do run pl_pf1_mid${m}_nch$n MF_SWEEP_PF=1 MF_SWEEP_MID=$m MF_SWEEP_MID_NCH=$n; done; done
run pl_pf1_noskew MF_SWEEP_PF=1 MF_SWEEP_SKEW=0
run pl_pf0_noskew MF_SWEEP_SKEW=0
run pl_db16_noskew MF_SWEEP_DB=1 MF_SWEEP_SKEW=0
run pl_db32_noskew MF_SWEEP_DB=1 MF_SWEEP_DB_NCH=32 MF_SWEEP_SKEW=0
SKEW=
run un_pf0
run un_pf1 MF_SWEEP_PF=1
run un_db16 MF_SWEEP_DB=1
CFG=cfg4 STEPS=10 WARM=2 CHECK=
run cfg4_pf0
run cfg4_pf1 MF_SWEEP_PF=1
run cfg4_pf0_b
run cfg4_pf1_b MF_SWEEP_PF=1
