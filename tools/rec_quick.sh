#!/bin/bash
# recommend step alone at a few K (131072 users x 100000 items): TFLOP/s of the whole step
for k in 100 30 64 128 256; do
  echo "K=$k: $(python tools/rec_bench.py --users 131072 --items 100000 --feats $k --reps 2 | tail -1)"
done
