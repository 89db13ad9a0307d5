#!/bin/bash
# round 3: the GPU suite twice (rare mismatches show as flaky failures), then the lockstep hunt on the new ordered-sum loop
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R
for i in 1 2; do python -m pytest tests -q -m gpu > $O/gpu_tests_$i.log 2>&1; tail -2 $O/gpu_tests_$i.log; grep -n "^FAILED\|AssertionError" $O/gpu_tests_$i.log | head -5; done
DBG_CFG=nflx10 DBG_ITERS=1200 timeout -k 10 400 python3 tools/skew_dbg.py MF_SWEEP_LONG=3000 > $O/dbg_nflx10.txt 2>&1; tail -2 $O/dbg_nflx10.txt
DBG_ITERS=1500 timeout -k 10 300 python3 tools/skew_dbg.py - MF_SWEEP_LONG=400 > $O/dbg_cfg3.txt 2>&1; tail -3 $O/dbg_cfg3.txt
