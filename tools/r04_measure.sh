#!/bin/bash
# Round-4 measurement pass on the GPU box: every BASELINE workload with a bench line + rocprofv3 summaries.
# usage: tools/r04_measure.sh <tag> [what...]   what in: tests default c3 hoh cheb mix fcc imp sq   (default: all but tests)
set -e
TAG=${1:-r04}; shift || true
WHAT=${*:-default c2 hoh cheb mix fcc imp kubo sq}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd $ROOT
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }; tail -2 $O/pytest_gpu.log; fi
run_bench() { # name args...
  local name=$1; shift
  echo "== bench $name: $*"
  python3 bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -20 $O/bench_$name.err; exit 1; }
  python3 - $O/bench_$name.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]
print("   %s: %.1f TFLOP/s  %.1f ms/step  %.1f sites/s  frac_kernel %.3f (algorithmic %.3f, executed %.3f) frac_step %.3f (algorithmic %.3f) hop %.3f ms  cpu %s" % (d["config"]["workload_key"], d["value"]*1e-3, d["ms_per_step"], d.get("sites_per_s", d.get("vectors_per_s", 0.0)), r["frac_kernel"], r["frac_algorithmic"], r.get("executed",{}).get("frac",0), r["frac_step"], r["frac_step_algorithmic"], r["avg_launch_ms"], d.get("cpu_baseline",{}).get("value")))
PY
}
prof() { # tag key extra...
  local t=$1 k=$2; shift 2
  PROFILE_NAME=${TAG}_${t}_rocprof_summary.txt BENCH_EXTRA="$*" tools/profile_bench.sh ${TAG}_$t $k > $O/prof_$t.log 2>&1 || { tail -20 $O/prof_$t.log; exit 1; }
}
# (round 4: bench.py's default IS the 46^3 north-star cell; the 22^3 variants name their cell)
if has default; then run_bench default --steps 20 --warmup 5; prof c3 block_c46_s64_l50; fi
if has c2; then run_bench c2 --cells 22 --steps 5 --warmup 1; prof c2 block_c22_s64_l50 --cells 22; fi
if has hoh; then run_bench hoh --cells 22 --hoh --steps 3 --warmup 1; prof hoh block_hoh_c22_s64_l50 --cells 22 --hoh; fi
if has cheb; then run_bench cheb --cells 22 --recur chebyshev --steps 3 --warmup 1; prof cheb chebyshev_c22_s64_l50 --cells 22 --recur chebyshev; fi
if has mix; then run_bench mix --cells 22 --spin-mixing --steps 3 --warmup 1; prof mix block_mix_c22_s64_l50 --cells 22 --spin-mixing; fi
if has fcc; then run_bench fcc --workload fccCu001 --steps 3 --warmup 1; prof fcc chebyshev_fccCu001_s64_l50 --workload fccCu001; fi
if has imp; then run_bench imp --workload B2FeCo --steps 3 --warmup 1; prof imp block_hoh_B2FeCo_s64_l50 --workload B2FeCo; fi
if has kubo; then run_bench kubo50 --workload kubo --cond-ll 50 --steps 3 --warmup 1; run_bench kubo500 --workload kubo --cond-ll 500 --steps 1 --warmup 1; prof kubo kubo_c20_l50 --workload kubo --cond-ll 50; fi
if has sq; then tools/pmc_sq.sh ${TAG}_c2 --cells 22 > $O/sq_c2.txt 2>&1 || { tail $O/sq_c2.txt; exit 1; }; tools/pmc_sq.sh ${TAG}_mix --cells 22 --spin-mixing > $O/sq_mix.txt 2>&1 || { tail $O/sq_mix.txt; exit 1; }; fi
echo done
