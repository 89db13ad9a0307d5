#!/bin/bash
# round 3: A/B of the mid-length rows' own launch (double-buffered form, large chunk) on the cfg3 power-law shape
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3c; mkdir -p $O; cd $R
export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
run() { name=$1; shift; env "$@" python3 bench.py --config ${CFG:-cfg3} $SKEW --steps ${STEPS:-200} --warmup ${WARM:-20} --no-cpu-baseline --no-recommend --check > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; }
  python3 - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    k = r["kernel"]
    print("%-30s ms %8.4f frac %.3f item %.4f user %.4f bit-identical %s %s | long%s mid%s" % (sys.argv[2], d["ms_per_step"], r["frac"], r.get("item_sweep_ms") or 0, r.get("user_sweep_ms") or 0,
          d["check"]["L_bit_identical"], d["check"]["R_bit_identical"], k.split("long_rows")[1].split()[0], k.split("mid_rows")[1].split()[0]))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
SKEW=--skew
run mid_off MF_SWEEP_MID=0
run mid_rule
for m in 128 192 256 384; do for n in 24 32 48 64; do run mid${m}_nch$n MF_SWEEP_MID=$m MF_SWEEP_MID_NCH=$n; done; done
for t in 1200 2000 3000; do for m in 192 256; do run long${t}_mid${m}_nch48 MF_SWEEP_LONG=$t MF_SWEEP_MID=$m MF_SWEEP_MID_NCH=48; done; done
