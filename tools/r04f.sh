#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r04f; mkdir -p $O
cd $ROOT
python -m pytest tests/test_gpu_kubo.py "tests/test_gpu_orbital.py::test_zero_edit_program_reproduces_the_reference_file_on_a_magnetic_case" -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --workload kubo --cond-ll 50 --steps 3 --warmup 1 --no-cpu > $O/kubo50.json 2> /dev/null
python bench.py --workload kubo --cond-ll 500 --steps 1 --warmup 1 --no-cpu > $O/kubo500.json 2> /dev/null
python - <<'PY'
import json
for f in ("kubo50", "kubo500"):
    try:
        d = json.loads([l for l in open("gpurun_out/r04f/%s.json" % f) if l.startswith("{")][-1]); g = d["roofline"]["gemm"]
        print(f, "ms/step %.1f dev %.1f" % (d["ms_per_step"], d["device_ms_per_step"]), "value %.1f TF" % (d["value"] * 1e-3), "gemm ms %.1f  TF %.1f  share %.2f" % (g["ms_per_step"], g["achieved"], g["share_of_device_time"]), "spmm frac %.3f" % d["roofline"]["frac"])
    except Exception as e:
        print(f, "failed", e)
PY
