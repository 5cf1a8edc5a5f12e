#!/bin/bash
# Builds the Fortran host side against the compiled reference modules (oracle/_ref, build container only):
#   oracle/_ref/obj/{rsrec_binding,recursion_gpu,green_gpu,bands_gpu,hamiltonian_gpu}.o   the binding + the drop-in types
#   oracle/_ref/rslmto_gpu.x                          reference workflow + GPU recursion (end-to-end drop-in test binary; program: tests/fortran/)
#   oracle/_ref/kubo_gpu.x                            reference conductivity post-processing + GPU Kubo moments
#   oracle/_ref/nncal_check.x                         reference pre-processing with type(lattice) or type(lattice_cells) (CPU only)
# The outputs live under oracle/_ref because they contain reference object code (git-ignored, travels to the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
OUT="$ROOT/oracle/_ref"
FC="${FC:-/opt/rocm/bin/amdflang}"
MKLDIR="${MKLDIR:-/opt/conda/lib}"
if [ ! -f "$OUT/librslmto_ref.a" ]; then echo "oracle/_ref not built: skipping Fortran drop-in build"; exit 0; fi
FFLAGS="-cpp -O2 -fopenmp -J$OUT/mod -I$OUT/mod"
for f in rsrec_binding rsrec_context recursion_gpu dos_gpu green_gpu bands_gpu hamiltonian_gpu lattice_cells; do
  (cd "$OUT/obj" && "$FC" $FFLAGS -c "$HERE/$f.f90" -o "$OUT/obj/$f.o")
done
# the test programs (tests/fortran/): the reference's workflows with the drop-in types behind them
for f in scf_gpu_driver kubo_gpu_driver nncal_check; do
  (cd "$OUT/obj" && "$FC" $FFLAGS -c "$ROOT/tests/fortran/$f.f90" -o "$OUT/obj/$f.o")
done
"$FC" "$OUT/obj/nncal_check.o" "$OUT/obj/lattice_cells.o" "$OUT/librslmto_ref.a" -fopenmp -L"$MKLDIR" -lmkl_rt -Wl,-rpath,"$MKLDIR" -o "$OUT/nncal_check.x"
"$FC" "$OUT/obj/scf_gpu_driver.o" "$OUT/obj/bands_gpu.o" "$OUT/obj/green_gpu.o" "$OUT/obj/hamiltonian_gpu.o" "$OUT/obj/recursion_gpu.o" "$OUT/obj/lattice_cells.o" "$OUT/obj/rsrec_context.o" "$OUT/obj/rsrec_binding.o" "$OUT/librslmto_ref.a" \
  -fopenmp -L"$MKLDIR" -lmkl_rt -Wl,-rpath,"$MKLDIR" \
  -L"$ROOT/rslmtoasa_amd" -lrsrec -Wl,-rpath,'$ORIGIN/../../rslmtoasa_amd' -Wl,-rpath,/opt/rocm/lib \
  -o "$OUT/rslmto_gpu.x"
"$FC" "$OUT/obj/kubo_gpu_driver.o" "$OUT/obj/recursion_gpu.o" "$OUT/obj/rsrec_context.o" "$OUT/obj/rsrec_binding.o" "$OUT/librslmto_ref.a" \
  -fopenmp -L"$MKLDIR" -lmkl_rt -Wl,-rpath,"$MKLDIR" \
  -L"$ROOT/rslmtoasa_amd" -lrsrec -Wl,-rpath,'$ORIGIN/../../rslmtoasa_amd' -Wl,-rpath,/opt/rocm/lib \
  -o "$OUT/kubo_gpu.x"
echo "built $OUT/rslmto_gpu.x $OUT/kubo_gpu.x"
