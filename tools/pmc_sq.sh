#!/bin/bash
# SQ issue/wait breakdown of the recursion kernels (one PMC pass; no tracing domains besides --kernel-trace).
# Usage (GPU box): tools/pmc_sq.sh <tag> [bench args]
set -e
TAG=${1:-sq}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
   --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-green "$@" > /dev/null 2> $OUT/a.log || { tail -5 $OUT/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU \
   --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-green "$@" > /dev/null 2> $OUT/b.log || { tail -5 $OUT/b.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:36]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:5]:
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %.5g" % (c, v))
PY
