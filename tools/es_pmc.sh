#!/bin/bash
# PMC counters of the streams launch on instML100k (debugging aid)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  MF_ITER_MODE=es rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/espmc_$tag -o out --output-format csv -- python3 bench.py --config ml100k --steps 50 --warmup 5 --no-cpu-baseline --no-recommend > /dev/null 2>&1
  f=$(find gpurun_out/espmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
for k,v in agg.items():
    if "stream" in k or "sweep_dma" in k:
        print(k, {a:int(b) for a,b in v.items()})
PY
done
