# Kernel timeline of one iteration of the Netflix-shaped power-law instance for several long-row thresholds.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for thr in ${THRS:-default}; do
  rm -rf $O/prof_nflx; cd /tmp
  if [ $thr = default ]; then unset MF_SWEEP_LONG; else export MF_SWEEP_LONG=$thr; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_nflx -- python3 $R/bench.py --config nflx --steps 6 --warmup 2 --no-cpu-baseline > $O/nflx_$thr.json 2>/dev/null || exit 1
  python3 - <<PY
import csv,glob,json
f=glob.glob("$O/prof_nflx/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
sw=[r for r in rows if "mf::" in r["Kernel_Name"] and ("sweep" in r["Kernel_Name"] or "ordered" in r["Kernel_Name"])]
j=json.load(open("$O/nflx_$thr.json"))
print("threshold $thr: ms/iter %.2f item %.2f user %.2f  %s" % (j["ms_per_step"], j["roofline"]["item_sweep_ms"], j["roofline"]["user_sweep_ms"], j["roofline"]["kernel"].split("lds=")[1]))
last=sw[-4:] if len(sw)>=4 else sw
t0=min(int(r["Start_Timestamp"]) for r in last)
for r in last: print("   ", r["Kernel_Name"][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else "", "%.2f -> %.2f ms" % ((int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6))
PY
done
