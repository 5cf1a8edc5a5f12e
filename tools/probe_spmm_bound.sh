#!/bin/bash
# Timing probes that separate "matrix pipe" from "operand delivery" in the SpMM kernels (results are WRONG by construction):
#   S5_PROBE / S4_PROBE = 1: no operator-fragment loads in the slot loop, 2: no psi loads, 3: neither.
# Builds the variant libraries into build/probe/ (build container or GPU box) and, on a GPU box, runs
# tools/probe_spmm_locality.py with each of them (RSREC_LIB selects the library).  Numbers quoted in DESIGN.md /
# kernels_spmm5.hpp come from this script.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/probe
cd $ROOT/rslmtoasa_amd/csrc
for P in 1 2 3; do
  if [ ! -f $ROOT/build/probe/librsrec_p$P.so ] || [ rsrec.hip -nt $ROOT/build/probe/librsrec_p$P.so ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-unused-function -DS5_PROBE=$P -DS4_PROBE=$P -shared rsrec.hip -o $ROOT/build/probe/librsrec_p$P.so
  fi
done
if python3 -c "import ctypes; ctypes.CDLL('$ROOT/rslmtoasa_amd/librsrec.so').rsrec_device_count() > 0 or exit(1)" 2>/dev/null; then
  cd $ROOT
  for P in 0 1 2 3; do
    if [ $P = 0 ]; then unset RSREC_LIB; else export RSREC_LIB=$ROOT/build/probe/librsrec_p$P.so; fi
    echo "probe $P"; python3 tools/probe_spmm_locality.py 2>&1 | tail -2
  done
fi
