# usage (GPU box): bash tools/pmc_traffic.sh <key> <bench.py args...>
# Two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over a short bench run; prints the per-launch HBM-side bytes
# of the sweep kernel = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts half).
key=$1; shift
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$key; rm -rf $O; mkdir -p $O; cd /tmp
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $O/write.json 2> $O/write.err || exit 1
python3 - <<PY
import csv, glob, json
O = "$O"
def pmc(name, ctr):
    f = glob.glob(O + "/%s/*/*counter_collection.csv" % name)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "mf::" in r["Kernel_Name"] and ("sweep" in r["Kernel_Name"] or "ordered_sum" in r["Kernel_Name"]) and r["Counter_Name"] == ctr:
            per.setdefault(r["Kernel_Name"][:48] + " grid=" + r["Grid_Size"], []).append(float(r["Counter_Value"]))
    return per
f, w = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
j = json.load(open(O + "/fetch.json"))
out = {"workload": j["config"]["workload"], "algorithmic_bytes_per_launch": j["roofline"]["algorithmic_bytes_per_launch"], "kernels": {}}
tot = 0.0
for k in sorted(f):
    b = (2 * sum(f[k]) / len(f[k]) + sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)) * 1024
    out["kernels"][k] = {"launches_sampled": len(f[k]), "hbm_bytes_per_launch": b}
    tot += b
out["hbm_bytes_per_iteration"] = tot
print(json.dumps(out, indent=1))
json.dump(out, open(O + "/summary.json", "w"), indent=1)
PY
