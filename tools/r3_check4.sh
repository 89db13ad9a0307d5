#!/bin/bash
# round 3: the GPU suite three times in a row (rare mismatches show as flaky failures)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3l; mkdir -p $O; cd $R
for i in 1 2 3; do python -m pytest tests -q -m gpu > $O/gpu_tests_$i.log 2>&1; tail -1 $O/gpu_tests_$i.log; grep -n "^FAILED" $O/gpu_tests_$i.log | head -5; done
