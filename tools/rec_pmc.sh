# PMC of the MFMA recommend kernel (tools/rec_bench.py, 131072 x 100000, K=100): MFMA pipe busy, LDS, waits.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rec_pmc; rm -rf $O; mkdir -p $O; cd /tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$O/$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $d -- python3 $R/tools/rec_bench.py --users 131072 --items 100000 --feats 100 --reps 1 > /dev/null 2>&1 || exit 1
done
python3 - <<PY
import csv, glob
tot = {}
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "recommend_mfma" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            tot["_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot["_name"] = r["Kernel_Name"]
print("#", tot.pop("_name"), "kernel %.3f ms" % (tot.pop("_ns") / 1e6))
cyc = tot["GRBM_GUI_ACTIVE"] / 8
print("# shader cycles (GRBM_GUI_ACTIVE / 8 XCDs) %.4g; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles = %.1f %%" % (cyc, 100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc))
print("# waves waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) = %.1f %%" % (100 * tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"]))
for k in sorted(tot): print(k, tot[k])
PY
