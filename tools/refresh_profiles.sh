# Runs on the GPU box: kernel-trace stats, the two PMC passes (separately, as the guide prescribes) and the bench
# line of the default configuration; raw outputs under gpurun_out/refresh/, summaries copied by the caller.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --recommend > $O/stats_bench.json 2> $O/stats.err || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
cd $R && python3 bench.py --recommend > $O/bench.json 2> $O/bench.err || exit 1
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
        if "sweep_dma_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            (item if int(r["Grid_Size"]) // 64 <= 100000 else user).append(float(r["Counter_Value"]))
    return item, user
stats()
fi, fu = pmc("pmc_fetch", "FETCH_SIZE"); wi, wu = pmc("pmc_write", "WRITE_SIZE")
avg = lambda v: sum(v) / max(len(v), 1)
ib = (2 * avg(fi) + avg(wi)) * 1024; ub = (2 * avg(fu) + avg(wu)) * 1024
print("item sweep launches", len(fi), len(wi), "bytes", ib, " user sweep launches", len(fu), len(wu), "bytes", ub)
json.dump({"item_sweep_bytes": ib, "user_sweep_bytes": ub, "hbm_bytes_per_launch": (ib + ub) / 2}, open(O + "/pmc_summary.json", "w"))
PY
cat $O/bench.json
