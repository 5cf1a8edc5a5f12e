#!/bin/bash
# A/B of two builds of the library on one bench workload with per-kernel times:  tools/ab_lib.sh "<bench args>" <libA.so> <libB.so> ...
# (variant builds: hipcc ... -D<MACRO>=... -o build/<name>.so; RSREC_LIB selects the library the ctypes mirror loads)
ARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for L in "$@"; do
    D=$ROOT/gpurun_out/ab_lib/$(basename $L .so)_$rep
    rm -rf $D; mkdir -p $D
    RSREC_LIB=$ROOT/$L rocprofv3 --kernel-trace --stats --output-format csv -d $D -o p -- python3 $ROOT/bench.py $ARGS --no-cpu --no-green --steps 3 --warmup 1 > $D/bench.log 2>&1 || { tail -5 $D/bench.log; exit 1; }
    python3 - "$D" "$L" <<'PY'
import csv, glob, json, sys
d, lib = sys.argv[1], sys.argv[2]
line = [l for l in open(d + "/bench.log") if l.startswith("{")][-1]
b = json.loads(line)
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:5]
print("%-28s %8.2f ms/step | " % (lib, b["ms_per_step"]) + " | ".join("%s %.3f ms" % (r["Name"].split("(")[0].replace("void rsrec::", "").replace("rsrec::", "")[:18], float(r["AverageNs"]) / 1e6) for r in rows))
PY
  done
done
