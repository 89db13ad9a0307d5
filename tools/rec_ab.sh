set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "recommend or certification or golden_full or scored" > gpurun_out/rec2_tests.log 2>&1 || { tail -30 gpurun_out/rec2_tests.log; exit 1; }
tail -2 gpurun_out/rec2_tests.log
for K in 100 64 30 128 60; do
  for ares in 1 0; do
    echo "K=$K ARES=$ares"
    MF_RECOMMEND_ARES=$ares python tools/rec_bench.py --users 1000000 --items 100000 --feats $K --reps 2 | tail -1
  done
done
