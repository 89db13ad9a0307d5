#!/bin/bash
# kernel durations of instML100k iterations under rocprofv3 (kernel trace): errors + resident streams, and the sweeps
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ml100k_trace; rm -rf $O; mkdir -p $O; cd /tmp
for mode in auto sweeps; do
  MF_ITER_MODE=$mode timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -- python3 $R/bench.py --config ml100k --steps 300 --warmup 50 --no-cpu-baseline --no-recommend > $O/$mode.json 2> $O/$mode.err || exit 1
  f=$(find $O/$mode -name "*kernel_stats.csv" | head -1)
  echo "== MF_ITER_MODE=$mode"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mf::" in r["Name"] and int(r["Calls"])>=100: print("%-70s calls %5s avg %8.0f ns min %6s max %6s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
  cp $f $O/${mode}_kernel_stats.csv
done
