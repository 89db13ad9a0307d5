# recommend parity tests, then timing of the MFMA recommend forms over K (A/B: resident L, LDS-DMA staging of R)
set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "recommend or certification or golden_full or scored" > gpurun_out/rec2_tests.log 2>&1 || { tail -30 gpurun_out/rec2_tests.log; exit 1; }
tail -2 gpurun_out/rec2_tests.log
for K in ${KS:-100 64 30 128 60}; do
  for mode in "1 1" "1 0" "0 0"; do
    set -- $mode
    echo "K=$K resident_L=$1 R_by_LDS_DMA=$2"
    MF_RECOMMEND_ARES=$1 MF_RECOMMEND_BDMA=$2 python tools/rec_bench.py --users 1000000 --items 100000 --feats $K --reps 2 2>/dev/null | tail -1
  done
done
