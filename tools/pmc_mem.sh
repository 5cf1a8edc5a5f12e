#!/bin/bash
# Texture-addresser / L1 / L2 utilisation of the recursion kernels (PMC passes; no tracing domains besides --kernel-trace).
# Usage (GPU box): tools/pmc_mem.sh <tag> [bench args]
set -e
TAG=${1:-mem}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcmem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-green $BENCH_ARGS > /dev/null 2> $OUT/$name.log || { tail -5 $OUT/$name.log; exit 1; }
}
BENCH_ARGS="$*"
# at most two counters of a block per pass (more: "Request exceeds the capabilities of the hardware to collect")
run a GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run b GRBM_GUI_ACTIVE TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run c GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run d GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum
run e GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_REQ_sum
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:36]
        acc[k][r["Counter_Name"] + "@" + f.split(os.sep)[-3]] += float(r["Counter_Value"])
for k, d in sorted(acc.items(), key=lambda kv: -max(kv[1].values()))[:4]:
    print(k)
    for c, v in sorted(d.items()):
        print("   %-44s %.5g" % (c, v))
PY
