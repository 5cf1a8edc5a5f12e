#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r04g; mkdir -p $O
cd $ROOT
python -m pytest tests/test_gpu_green.py tests/test_gpu_ldos.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python tools/time_green.py 2>&1 | tail -2
python tools/time_green.py 2>&1 | tail -2
