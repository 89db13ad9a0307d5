#!/bin/bash
# round 3: the mid-length rows through the row-cooperative kernel (8 waves per row, products in LDS), thresholds swept
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3e; mkdir -p $O; cd $R
export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
run() { name=$1; shift; env "$@" python3 bench.py --config ${CFG:-cfg3} $SKEW --steps ${STEPS:-200} --warmup ${WARM:-20} --no-cpu-baseline --no-recommend $CHECK > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; }
  python3 - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    k = r["kernel"]; c = d.get("check") or {}
    print("%-34s ms %8.4f frac %.3f item %.4f user %.4f bit-identical %s %s | long%s mid%s" % (sys.argv[2], d["ms_per_step"], r["frac"], r.get("item_sweep_ms") or 0, r.get("user_sweep_ms") or 0,
          c.get("L_bit_identical"), c.get("R_bit_identical"), k.split("long_rows")[1].split()[0], k.split("mid_rows")[1].split()[0]))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
CHECK=--check
SKEW=--skew
run base MF_SWEEP_MID=0
for m in 96 160 256 400; do for n in 8 13; do run coop_mid${m}_nch$n MF_SWEEP_MID_KERNEL=coop MF_SWEEP_MID=$m MF_SWEEP_MID_NCH=$n; done; done
for t in 1500 2500 4000; do for m in 128 256; do run coop_long${t}_mid${m} MF_SWEEP_MID_KERNEL=coop MF_SWEEP_LONG=$t MF_SWEEP_MID=$m MF_SWEEP_MID_NCH=13; done; done
