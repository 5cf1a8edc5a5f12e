#!/bin/bash
# PROBE: one pool for the three work vectors with spacing size + k * 64 KB: does the spacing decide the level-parity alternation of the kernels' times?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export RSREC_LIB=$ROOT/build/librsrec_probe.so
for k in 0 1 3 17 32 257 4099; do
  echo "== pool spacing + $k x 64 KB"
  RSREC_PROBE_ALLOC=$k $ROOT/tools/per_level_trace.sh pool$k "$@" | python3 -c "
import sys
for l in sys.stdin:
    p=l.split()
    if len(p)>40: print(p[0], ' '.join(p[-6:]))"
done
