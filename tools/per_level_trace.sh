#!/bin/bash
# Per-launch durations of the level kernels for one step (kernel trace), next to the active-atom count of every level.
# Usage (GPU box): tools/per_level_trace.sh <tag> [bench args]
TAG=${1:-lv}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lvtrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-green "$@" > $OUT/bench.json 2> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
python3 - $OUT "$@" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
f = glob.glob(os.path.join(out, "t", "*", "*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
by = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rsrec::", "").split("<")[0]
    by.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
for k in ("k_spmm5", "k_mfma_adot", "k_mfma_orth3"):
    if k in by:
        print(k, " ".join("%.2f" % v for v in by[k]))
PY
