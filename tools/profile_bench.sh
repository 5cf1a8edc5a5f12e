#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline object cites: a kernel-trace/stats pass and separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, TCC hit/miss -- one counter set per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage (on the GPU box): [BENCH_EXTRA="--cells 22"] tools/profile_bench.sh <tag> [workload_key]
#   -> gpurun_out/prof_<tag>/{stats,pmc_*}, gpurun_out/prof_<tag>/summary.txt, gpurun_out/prof_<tag>/traffic.json
set -e
TAG=${1:-run}
KEY=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu --no-green $BENCH_EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.log
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_$N.log
done
if [ -n "$KEY" ]; then
  PROFILE_NAME=${PROFILE_NAME:-r04_${TAG}_rocprof_summary.txt} python3 $ROOT/tools/summarize_pmc.py $OUT --traffic-key $KEY --traffic-out $OUT/traffic.json > $OUT/summary.txt
else
  python3 $ROOT/tools/summarize_pmc.py $OUT > $OUT/summary.txt
fi
cat $OUT/summary.txt
# raw csv trees are large: keep the summaries and the kernel stats only
find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*agent_info.csv' -delete
