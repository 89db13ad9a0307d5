#!/bin/bash
# round 3: whole GPU suite + the cfg3 / nflx lines with the shipped library (no environment switches)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
line() { python3 - $1 "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = d["roofline"]; c = d.get("check") or {}
    print("%-16s ms %9.4f frac %.3f item %.4f user %.4f bit-identical %s %s | %s" % (sys.argv[2], d["ms_per_step"], r["frac"], r.get("item_sweep_ms") or 0, r.get("user_sweep_ms") or 0, c.get("L_bit_identical"), c.get("R_bit_identical"), r["kernel"].split("lds=")[1][:150]))
except Exception as e: print(sys.argv[2], "no line", e)
PY
}
python3 bench.py --config cfg3 --skew --steps 200 --warmup 20 --no-cpu-baseline --check > $O/cfg3_powerlaw.json 2> $O/cfg3_powerlaw.err; line $O/cfg3_powerlaw.json cfg3_powerlaw
python3 bench.py --config cfg3 --steps 200 --warmup 20 --no-cpu-baseline --check > $O/cfg3_uniform.json 2> $O/cfg3_uniform.err; line $O/cfg3_uniform.json cfg3_uniform
python3 bench.py --config nflx --steps 20 --warmup 3 --no-cpu-baseline --check > $O/nflx.json 2> $O/nflx.err; line $O/nflx.json nflx
python3 bench.py --config ml100k --steps 3000 --warmup 100 --one-call --no-cpu-baseline > $O/ml100k_onecall.json 2> $O/ml100k.err; line $O/ml100k_onecall.json ml100k_onecall
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recommend > $O/cfg4.json 2> $O/cfg4.err; line $O/cfg4.json cfg4
