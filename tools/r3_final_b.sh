#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R
python3 bench.py > $O/cfg4_n1_bench.json 2> $O/cfg4_n1_bench.err; python3 - $O/cfg4_n1_bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]; c=d["cpu_baseline"]
print("default line: ms", d["ms_per_step"], "value %.4g"%d["value"], "frac", r["frac"], "traffic", r["traffic"], "rec", d["recommend"]["tflops"], "cpu %.3g cores %d x%.0f"%(c["value"], c["cores"], c["gpu_over_cpu"]))
PY
bash tools/pmc_traffic.sh cfg5_n1 --config cfg5 --no-recommend > $O/cfg5_pmc.txt 2>&1; tail -12 $O/cfg5_pmc.txt
bash tools/config_benches.sh 2>&1 | tail -12
