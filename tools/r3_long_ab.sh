#!/bin/bash
# round 3: the extreme-row threshold re-swept on the large skewed shapes now that a lone wave walks 0.13 us per entry
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3r; mkdir -p $O; cd $R
run() { name=$1; shift; env "$@" python3 bench.py $ARGS --no-cpu-baseline --no-recommend > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; }
  python3 - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
    k = r["kernel"]
    print("%-26s ms %8.3f frac %.3f item %.3f user %.3f | long%s" % (sys.argv[2], d["ms_per_step"], r["frac"], r.get("item_sweep_ms") or 0, r.get("user_sweep_ms") or 0, k.split("long_rows")[1].split()[0]))
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
ARGS="--config nflx --steps 20 --warmup 3"
run nflx_rule
for t in 40000 60000 90000 140000; do run nflx_long$t MF_SWEEP_LONG=$t; done
ARGS="--columns zipf --steps 20 --warmup 3"
run zipf_rule
for t in 80000 160000 320000; do run zipf_long$t MF_SWEEP_LONG=$t; done
