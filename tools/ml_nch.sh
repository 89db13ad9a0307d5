for n in default 8 20 28 32; do
  if [ $n = default ]; then unset MF_SWEEP_NCH; else export MF_SWEEP_NCH=$n; fi
  python bench.py --config ml100k --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null > gpurun_out/ml.json
  python -c "import json; j=json.load(open('gpurun_out/ml.json')); r=j['roofline']; print('nch=$n us/iter %.1f item %.1f user %.1f  %s' % (j['ms_per_step']*1e3, r['item_sweep_ms']*1e3, r['user_sweep_ms']*1e3, r['kernel'][-40:]))"
done
