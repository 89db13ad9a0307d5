# Runs on the GPU box: kernel-trace stats, the two PMC passes (separately, as the guide prescribes) and the bench
# line of the default configuration; raw outputs under gpurun_out/refresh/, summaries copied by the caller.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.err || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
cd $R && python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
python3 - <<PY
import csv, glob, json
O = "$O"
def stats():
    f = glob.glob(O + "/stats/*/*kernel_stats.csv")[0]
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if any(k in r["Name"] for k in ("sweep", "recommend", "ordered", "row_norm"))]
    with open(O + "/kernel_stats_summary.csv", "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
    for r in keep: print(r["Name"][:70], r["Calls"], "avg_ns", r["AverageNs"])
def pmc(name, ctr):
    f = glob.glob(O + "/%s/*/*counter_collection.csv" % name)[0]
    item, user = [], []
    for r in csv.DictReader(open(f)):
        # one workgroup per row: the item sweep has 1e5 of them (single waves or wave pairs), the user sweep 1e6
        if "mf::sweep_" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            (item if int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1) <= 100000 else user).append(float(r["Counter_Value"]))
    return item, user
stats()
fi, fu = pmc("pmc_fetch", "FETCH_SIZE"); wi, wu = pmc("pmc_write", "WRITE_SIZE")
avg = lambda v: sum(v) / max(len(v), 1)
ib = (2 * avg(fi) + avg(wi)) * 1024; ub = (2 * avg(fu) + avg(wu)) * 1024
print("item sweep launches", len(fi), len(wi), "bytes", ib, " user sweep launches", len(fu), len(wu), "bytes", ub)
import sys
sys.path.insert(0, "$R")
import recommender_system_amd as rs
bench = json.loads([l for l in open(O + "/bench.json") if l.startswith("{")][-1])
json.dump({"cfg4_n1": {"hbm_bytes_per_launch": (ib + ub) / 2, "item_sweep_bytes": ib, "user_sweep_bytes": ub,
                       "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
                       "kernel_source_hash": rs.capi.kernel_source_hash(),
                       "csvs": ["profiles/r03/cfg4_n1_pmc_FETCH_SIZE.csv", "profiles/r03/cfg4_n1_pmc_WRITE_SIZE.csv"],
                       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/refresh_profiles.sh); "
                                 "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts half of "
                                 "a 16-B/lane stream); L2->fabric requests, Infinity-Cache hits included"}},
          open(O + "/pmc_traffic_entry.json", "w"), indent=1)
PY
cat $O/bench.json
