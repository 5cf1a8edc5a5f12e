#!/bin/bash
# PROBE (needs a library built with the RSREC_PROBE_ORDER hook of DESIGN.md section 7): does the order in which the work vectors are allocated decide the
# rates of the post-hop passes' buffer arrangements?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export RSREC_LIB=$ROOT/build/librsrec_probe.so
for o in 0123 2013 0132 3210 1203 0123; do
  echo "== allocation order $o"
  RSREC_PROBE_ORDER=$o $ROOT/tools/per_level_trace.sh ord$o "$@" | python3 -c "
import sys
for l in sys.stdin:
    p=l.split()
    if len(p)>40 and p[0]!='k_spmm5': print(p[0], ' '.join(p[-6:]), ' sum %.1f' % sum(float(x) for x in p[1:]))"
done
