#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r04h; mkdir -p $O
cd $ROOT
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
tools/per_level_trace.sh c22b --cells 22 | tail -1 | cut -c1-400
python bench.py --cells 22 --steps 5 --warmup 1 --no-cpu > $O/c22.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu > $O/c46.json 2>/dev/null
python bench.py --workload fccCu001 --steps 3 --warmup 1 --no-cpu > $O/fcc.json 2>/dev/null
python bench.py --workload B2FeCo --steps 3 --warmup 1 --no-cpu > $O/imp.json 2>/dev/null
python bench.py --cells 22 --hoh --steps 3 --warmup 1 --no-cpu > $O/hoh.json 2>/dev/null
python bench.py --cells 22 --recur chebyshev --steps 3 --warmup 1 --no-cpu > $O/cheb.json 2>/dev/null
python - <<'PY'
import json
for f in ("c22","c46","fcc","imp","hoh","cheb"):
    try:
        d=json.loads([l for l in open("gpurun_out/r04h/%s.json"%f) if l.startswith("{")][-1]); r=d["roofline"]
        print(f, "%.1f ms/step  frac %.3f step %.3f  launch %.3f ms  host %.2f ms" % (d["ms_per_step"], r["frac"], r["frac_step"], r["avg_launch_ms"], d["host_ms_per_step"]))
    except Exception as e: print(f, "failed", e)
PY
