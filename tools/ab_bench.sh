#!/bin/bash
# A/B of library options on a bench workload: tools/ab_bench.sh "<bench args>" "<opt set A>" "<opt set B>" ...   (opt set: "k=v k=v", "" = defaults)
# Alternates the variants twice (box-to-box and thermal drift show up as spread between repeats of the same variant).
ARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
  for V in "$@"; do
    O=""; for kv in $V; do O="$O --opt $kv"; done
    python3 $ROOT/bench.py $ARGS --no-cpu --no-green --steps 3 --warmup 1 $O 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-34s %8.2f ms/step  hop %.3f ms  frac %.3f  step %.3f' % ('[$V]', d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_step']))"
  done
done
