#!/bin/bash
# persistent k_spmm5 on collinear operators: both output spins on every XCD (default since the end of round 3) against one spin per XCD
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for W in "" "--hoh" "--recur chebyshev" "--cells 46" "--cells 16"; do
  for O in "" "--opt s5_spin_xcd=1"; do
    python3 bench.py $W $O --no-cpu --no-green --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-22s %-22s %8.2f ms/step  hop %.3f ms  frac %.3f  step %.3f' % ('[$W]', '[$O]', d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_step']))"
  done
done
