#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=gpurun_out/r04l; mkdir -p $O
python -m pytest tests/test_gpu_kubo.py -x -q -m gpu > $O/kubo_tests.log 2>&1; echo "kubo tests rc=$?"; tail -3 $O/kubo_tests.log
for v in 1 4 8; do
  python bench.py --workload kubo --vectors $v --steps 2 --warmup 1 --no-cpu > $O/kubo50_v$v.json 2>/dev/null
  python - $O/kubo50_v$v.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("kubo cond_ll 50 vectors %d: %.2f ms/vector  spmm launch %.3f ms frac %.3f  gemm %.1f TF  value %.1f TF" % (d["config"]["vectors_per_step"], d["ms_per_vector"], r["avg_launch_ms"], r["frac"], r["gemm"]["achieved"], d["value"]*1e-3))
PY
done
python bench.py --workload kubo --vectors 8 --hoh --steps 1 --warmup 1 --no-cpu > $O/kubo50h_v8.json 2>/dev/null; python bench.py --workload kubo --vectors 1 --hoh --steps 2 --warmup 1 --no-cpu > $O/kubo50h_v1.json 2>/dev/null
python - $O <<'PY'
import json,sys
for f in ("kubo50h_v1","kubo50h_v8"):
    d=json.loads([l for l in open(sys.argv[1]+"/"+f+".json") if l.startswith("{")][-1]); r=d["roofline"]
    print(f, "%.2f ms/vector  spmm launch %.3f ms frac %.3f" % (d["ms_per_vector"], r["avg_launch_ms"], r["frac"]))
PY
echo "single site default"; python tools/time_single_site.py 1 | tail -2
echo "single site graph=0"; python tools/time_single_site.py 1 graph=0 | tail -2
echo "single site split 12"; python tools/time_single_site.py 1 graph=0 s5_split=3 s5_waves=12 s5_queue=2 | tail -2
echo "single site split 8"; python tools/time_single_site.py 1 graph=0 s5_split=3 s5_waves=8 s5_queue=2 | tail -2
echo "single site queue=2"; python tools/time_single_site.py 1 graph=0 s5_queue=2 | tail -2
BENCH_EXTRA="--cells 22 --opt s5_split=3 --opt s5_waves=12" tools/profile_bench.sh r04l_split12 > $O/prof_split12.log 2>&1; grep -i -A3 "k_spmm5" gpurun_out/prof_r04l_split12/summary.txt | head -30
