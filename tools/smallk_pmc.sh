#!/bin/bash
# HBM-side traffic of the sweeps at K = 10 and 30 on the cfg4 shape, with and without the 128-byte row pitch
# (two separate PMC passes each, tools/pmc_traffic.sh)
for k in 10 30; do for p in 1 0; do
  echo "== K=$k MF_ROW_PITCH=$p"
  MF_ROW_PITCH=$p bash tools/pmc_traffic.sh k${k}_p$p --config cfg4 --feats $k --no-recommend 2>&1 | python3 -c "
import json,sys
d=json.load(sys.stdin); a=d['algorithmic_bytes_per_launch']
for k,v in d['kernels'].items(): print('  %-60s %.3e B per launch = %.2f x algorithmic' % (k, v['hbm_bytes_per_launch'], v['hbm_bytes_per_launch']/a))"
done; done
