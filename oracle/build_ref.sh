#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- never part of the product path.
#
# Compiles the reference RS-LMTO-ASA Fortran sources *where they lie* under
# /root/reference/source (nothing is copied into this repo) with amdflang + MKL
# and leaves objects/modules/binaries in oracle/_ref/ (git-ignored).
# Recipe follows SURVEY.md section 8(c): module order derived from the `use`
# graph; the only source incompatibility with flang is GNU `zexp` (math.f90:584),
# handled with -Dzexp=exp.  The reference's own CMake build is NOT used.
#
# Outputs:
#   oracle/_ref/obj/*.o, oracle/_ref/mod/*.mod   reference modules
#   oracle/_ref/librslmto_ref.a                  all reference modules (no main)
#   oracle/_ref/rslmto_ref.x                     the reference program itself
#   oracle/_ref/dump_fixture.x                   our driver: full pipeline -> fixture
#   oracle/_ref/ref_kernel.x                     our driver: fixture in -> reference recursion -> outputs
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${RSREC_REFERENCE:-/root/reference}"
SRC="$REF/source"
OUT="$HERE/_ref"
FC="${FC:-/opt/rocm/bin/amdflang}"
MKLDIR="${MKLDIR:-/opt/conda/lib}"
if [ ! -d "$SRC" ]; then echo "reference sources not present ($SRC): skipping oracle/_ref build"; exit 0; fi
mkdir -p "$OUT/obj" "$OUT/mod"
FFLAGS="-cpp -O2 -fopenmp -DOpenMP_Fortran_FOUND -DCOLOR -Dzexp=exp -Dcdexp=exp -I$SRC -I$SRC/include_codes -J$OUT/mod -I$OUT/mod"
ORDER="face.F90 precision.f90 string.f90 logger.f90 array.f90 math.f90 namelist_generator.f90 control.f90 mpi.f90 globals.f90 report.f90 safe_alloc.f90 os.f90 element.f90 potential.f90 symbolic_atom.f90 lattice.f90 energy.f90 charge.f90 timer.f90 hamiltonian.f90 recursion.f90 density_of_states.f90 green.f90 bands.f90 xc.f90 mix.f90 self.f90 exchange.f90 conductivity.f90 include_codes/abspinlib/stdtypes.f90 include_codes/abspinlib/mtprng.f90 include_codes/abspinlib/parameters.f90 include_codes/abspinlib/constants.f90 include_codes/abspinlib/randomnumbers.f90 include_codes/abspinlib/depondt.f90 spin_dynamics.f90 calculation.f90 include_codes/abspinlib/abSpinlib.f90 include_codes/abspinlib/constrain.f90"
OBJS=""
for f in $ORDER; do
  o="$OUT/obj/$(basename "${f%.*}").o"
  if [ ! -f "$o" ] || [ "$SRC/$f" -nt "$o" ]; then
    echo "FC $f"
    (cd "$OUT/obj" && "$FC" $FFLAGS -c "$SRC/$f" -o "$o")
  fi
  OBJS="$OBJS $o"
done
rm -f "$OUT/librslmto_ref.a"
ar rcs "$OUT/librslmto_ref.a" $OBJS
LDFLAGS="-fopenmp -L$MKLDIR -lmkl_rt -Wl,-rpath,$MKLDIR"
(cd "$OUT/obj" && "$FC" $FFLAGS -c "$SRC/main.f90" -o "$OUT/obj/main.o")
"$FC" "$OUT/obj/main.o" "$OUT/librslmto_ref.a" $LDFLAGS -o "$OUT/rslmto_ref.x"
for drv in dump_fixture ref_kernel dump_kubo; do
  if [ -f "$HERE/$drv.f90" ]; then
    (cd "$OUT/obj" && "$FC" $FFLAGS -c "$HERE/$drv.f90" -o "$OUT/obj/$drv.o")
    "$FC" "$OUT/obj/$drv.o" "$OUT/librslmto_ref.a" $LDFLAGS -o "$OUT/$drv.x"
  fi
done
echo "oracle/_ref built"
