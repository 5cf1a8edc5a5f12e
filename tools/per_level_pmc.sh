#!/bin/bash
# Per-launch L2 hit rate and fetched bytes of k_spmm5 (one PMC pass each), in launch order.  Usage (GPU box): tools/per_level_pmc.sh <tag> [bench args]
TAG=${1:-lv}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lvpmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-green "$@" > /dev/null 2> $OUT/a.log || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-green "$@" > /dev/null 2> $OUT/b.log || { tail -5 $OUT/b.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
def rows(sub):
    f = glob.glob(os.path.join(out, sub, "*", "*counter_collection.csv"))[0]
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_spmm5" not in r["Kernel_Name"]:
            continue
        d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [d[k] for k in sorted(d)]
a, b = rows("a"), rows("b")
print("L2 hit rate per launch:", " ".join("%.2f" % (x["TCC_HIT_sum"] / max(x["TCC_HIT_sum"] + x["TCC_MISS_sum"], 1)) for x in a))
print("fetch GB per launch (2 x FETCH_SIZE):", " ".join("%.2f" % (2 * x["FETCH_SIZE"] * 1024e-9) for x in b))
PY
find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*kernel_trace.csv' -delete
