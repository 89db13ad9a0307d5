# kernel timeline of one iteration of the power-law cfg3 instance (two sweeps, extreme rows on the side stream);
# environment switches pass through (MF_HIP_LIB, MF_SWEEP_*); TRACE_OUT names the summary file
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; rm -rf $O/prof_skew; cd /tmp
MF_ITER_MODE=sweeps timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_skew -- python3 $R/bench.py --config cfg3 --skew --steps 20 --warmup 2 --no-cpu-baseline --no-recommend > /dev/null 2>&1
python3 - <<PY | tee ${TRACE_OUT:-$O/trace_skew.txt}
import csv,glob
f=glob.glob("$O/prof_skew/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
sw=[r for r in rows if "sweep" in r["Kernel_Name"] or "ordered" in r["Kernel_Name"]]
sw.sort(key=lambda r:int(r["Start_Timestamp"]))
last=sw[-12:]
t0=int(last[0]["Start_Timestamp"])
for r in last: print("%-58s grid %7s lds %6s  start %8.1f us  end %8.1f us  dur %7.1f" % (r["Kernel_Name"][:58], r["Grid_Size_X"], r.get("LDS_Block_Size",""), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
