#!/bin/bash
# bench lines of the other BASELINE-shaped configurations (one JSON line each under gpurun_out/configs/), every one
# that has a plain single-wave counterpart with --check (bit-identity of the factors against that form)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/configs; rm -rf $O; mkdir -p $O; cd $R
b() { name=$1; shift; python3 bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -3 $O/$name.err; return 1; }
      python3 - $O/$name.json $name <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]
rec = d.get("recommend") or {}; c = d.get("check") or {}
print("%-18s ms_per_step %9.4f  value %.3e  frac %.3f  item %.4f user %.4f  recommend %-22s check %s" % (sys.argv[2], d["ms_per_step"], d["value"], r["frac"], r.get("item_sweep_ms") or 0, r.get("user_sweep_ms") or 0,
      ("%.1f TFLOP/s %.3f" % (rec["tflops"], rec["frac_of_peak"])) if rec else "-", ("L %s R %s" % (c.get("L_bit_identical"), c.get("R_bit_identical"))) if c else "-"))
PY
}
b cfg3_uniform --config cfg3 --steps 200 --warmup 20 --check &&
b cfg3_powerlaw --config cfg3 --skew --steps 200 --warmup 20 --check &&
b cfg3_stratified --config cfg3 --columns stratified --steps 200 --warmup 20 --check &&
b ml100k_steps --config ml100k --steps 1000 --warmup 100 &&
b ml100k_onecall --config ml100k --steps 3000 --warmup 100 --one-call &&
b nflx --config nflx --steps 20 --warmup 3 --check &&
b cfg4_uniform --steps 20 --warmup 3 --check &&
b cfg4_stratified --columns stratified --steps 20 --warmup 3 --check &&
b cfg4_zipf --columns zipf --steps 20 --warmup 3 --check &&
b cfg5 --config cfg5 --steps 5 --warmup 1
