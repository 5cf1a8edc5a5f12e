cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for W in "--workload fccCu001 --recur chebyshev" "--workload B2FeCo --hoh"; do
  for P in "" "--no-positions"; do
    python3 bench.py $W $P --no-cpu --no-green --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-50s %8.2f ms/step  hop %.3f ms  frac %.3f  step %.3f' % ('[$W $P]', d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_step']))"
  done
done
done
