#!/bin/bash
# round 3: the pipelined closed-statement ordered sums: probe (time + serial-sum check), lockstep hunts, extreme-path tests
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R
(cd tools/micro && ./osum_probe item > $O/osum_item.txt 2>&1; ./osum_probe user > $O/osum_user.txt 2>&1; ./osum_probe item 2000 > $O/osum_item_stress.txt 2>&1)
head -6 $O/osum_item.txt; head -6 $O/osum_user.txt; tail -2 $O/osum_item_stress.txt
DBG_CFG=nflx10 DBG_ITERS=1500 timeout -k 10 400 python3 tools/skew_dbg.py MF_SWEEP_LONG=3000 > $O/dbg_nflx10.txt 2>&1; tail -1 $O/dbg_nflx10.txt
DBG_ITERS=1500 timeout -k 10 300 python3 tools/skew_dbg.py - MF_SWEEP_LONG=400 > $O/dbg_cfg3.txt 2>&1; tail -2 $O/dbg_cfg3.txt
python -m pytest tests -x -q -m gpu -k "extreme or skew or power_law or cfg3 or randomised or ordered" > $O/gpu_tests_subset.log 2>&1; tail -2 $O/gpu_tests_subset.log
