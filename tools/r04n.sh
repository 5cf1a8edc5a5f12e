#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=gpurun_out/r04n; mkdir -p $O
python -m pytest tests/test_gpu_density.py "tests/test_fortran_dropin.py" tests/test_gpu_spmm_random.py -x -q -m gpu -k "density or lanczos or split" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
tools/ab_bench.sh "--workload B2FeCo" "" "s5_octet=8" "s5_octet=1"
