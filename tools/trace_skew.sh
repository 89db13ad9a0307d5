export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; 
for k in 30 100; do rm -rf $O/prof_skew; cd /tmp; MF_SWEEP_LONG=128 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_skew -- python3 $R/bench.py --config cfg3 --feats $k --skew --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; python3 - <<PY
import csv,glob,statistics
f=glob.glob("$O/prof_skew/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
sw=[r for r in rows if "sweep" in r["Kernel_Name"] or "ordered" in r["Kernel_Name"]]
t0=int(sw[-6]["Start_Timestamp"])
print("K=$k")
for r in sw[-6:]: print(r["Kernel_Name"][:45], r["Grid_Size_X"], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3)
PY
done
