#!/bin/bash
# round 3, first GPU call: the whole GPU suite after the config / ordered-sum / multi rewrite, the cfg3 power-law line
# (with --check) as this round's starting point, and the ordered-sum probe (DPP and plain forms)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/r3a; mkdir -p $O; cd $R
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python3 bench.py --config cfg3 --skew --steps 200 --warmup 20 --no-cpu-baseline --check > $O/cfg3_powerlaw.json 2> $O/cfg3_powerlaw.err; tail -c 600 $O/cfg3_powerlaw.json
(cd tools/micro && ./osum_probe item > $O/osum_item.txt 2>&1; ./osum_probe user > $O/osum_user.txt 2>&1; OSUM_PLAIN=1 ./osum_probe item > $O/osum_item_plain.txt 2>&1)
tail -3 $O/osum_item.txt $O/osum_user.txt $O/osum_item_plain.txt
