gunzip -c tests/golden/instML100k.in.gz > /tmp/ml.in
for g in 2e6 2e7; do for rep in 1 2 3; do
  t0=$(date +%s.%N)
  MF_GRAPH_MAX=$g MATFACT_TIMING=1 ./recommender-system_amd/host/matFact /tmp/ml.in > /tmp/ml.out 2> /tmp/ml.err
  t1=$(date +%s.%N)
  echo "MF_GRAPH_MAX=$g wall $(python3 -c "print('%.3f' % ($t1 - $t0))") s | $(tail -1 /tmp/ml.err)"
done; done
cmp tests/golden/instML100k.out /tmp/ml.out && echo "output identical"
