#!/bin/bash
# ZERO-EDIT DROP-IN BUILD: the reference program linked from its own UNMODIFIED sources (main.f90, calculation.f90, self.f90 and every
# other file, compiled where they lie under /root/reference/source) with the GPU types behind the reference's module names.
#
# How: the six reference files whose types have a GPU counterpart are compiled under another module name
# (-D<name>_mod=<name>_ref_mod; the reference's sources are compiled with -cpp anyway, CMakeLists.txt), the GPU type of fortran/
# extends the reference type from there (same -D, so its `use <name>_mod` finds the renamed reference module), and
# fortran/shadow/<name>_mod.f90 re-exports the extended type under the reference's names.  Everything downstream -- `type(recursion) ::
# recursion_obj ; recursion_obj = recursion(hamiltonian_obj, energy_obj)` in calculation.f90:270,599, the `type(green)` dummy of
# bands' constructor (bands.f90:121), `type(bands)` in self's (self.f90:262) -- then declares, constructs and passes the GPU types.
# A maintainer's version of this recipe is a CMake diff of a dozen lines.
#
# Outputs (oracle/_ref/dropin/, git-ignored like the rest of oracle/_ref: they contain reference object code):
#   oracle/_ref/rslmto_dropin.x     = the reference's main program + librsrec behind its recursion / green / bands / hamiltonian / lattice modules
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
REF="${RSREC_REFERENCE:-/root/reference}"
SRC="$REF/source"
OUT="$ROOT/oracle/_ref/dropin"
FC="${FC:-/opt/rocm/bin/amdflang}"
MKLDIR="${MKLDIR:-/opt/conda/lib}"
if [ ! -d "$SRC" ]; then echo "reference sources not present ($SRC): skipping the zero-edit drop-in build"; exit 0; fi
mkdir -p "$OUT/obj" "$OUT/mod"
FFLAGS="-cpp -O2 -fopenmp -DOpenMP_Fortran_FOUND -DCOLOR -Dzexp=exp -Dcdexp=exp -I$SRC -I$SRC/include_codes -J$OUT/mod -I$OUT/mod"
OBJS=""
# compile <source> <object name> [extra flags]: rebuilt when the source is newer than the object
fc() {
  local src=$1 o="$OUT/obj/$2.o"; shift 2
  if [ ! -f "$o" ] || [ "$src" -nt "$o" ] || [ "$0" -nt "$o" ]; then
    echo "FC $(basename "$src") $*"
    (cd "$OUT/obj" && "$FC" $FFLAGS "$@" -c "$src" -o "$o")
  fi
  OBJS="$OBJS $o"
}
ref() { for f in "$@"; do fc "$SRC/$f" "$(basename "${f%.*}")"; done; }
# reference module <name> under the name <name>_ref_mod, the GPU type extending it, the shadow module handing it out as <name>_mod
shadowed() {
  local name=$1 gpu=$2
  fc "$SRC/$name.f90" "${name}_ref" "-D${name}_mod=${name}_ref_mod"
  fc "$HERE/$gpu.f90" "$gpu" "-D${name}_mod=${name}_ref_mod"
  fc "$HERE/shadow/${name}_mod.f90" "${name}_shadow"
}
ref face.F90 precision.f90 string.f90 logger.f90 array.f90 math.f90 namelist_generator.f90 control.f90 mpi.f90 globals.f90 report.f90 safe_alloc.f90 os.f90 element.f90 potential.f90 symbolic_atom.f90
shadowed lattice lattice_cells
ref energy.f90 charge.f90 timer.f90
fc "$HERE/rsrec_binding.f90" rsrec_binding
fc "$HERE/rsrec_context.f90" rsrec_context
shadowed hamiltonian hamiltonian_gpu
shadowed recursion recursion_gpu
shadowed density_of_states dos_gpu
shadowed green green_gpu
shadowed bands bands_gpu
ref xc.f90 mix.f90 self.f90 exchange.f90 conductivity.f90 include_codes/abspinlib/stdtypes.f90 include_codes/abspinlib/mtprng.f90 include_codes/abspinlib/parameters.f90 include_codes/abspinlib/constants.f90 include_codes/abspinlib/randomnumbers.f90 include_codes/abspinlib/depondt.f90 spin_dynamics.f90 calculation.f90 include_codes/abspinlib/abSpinlib.f90 include_codes/abspinlib/constrain.f90
LIBOBJS="$OBJS"
ref main.f90
"$FC" "$OUT/obj/main.o" $LIBOBJS -fopenmp -L"$MKLDIR" -lmkl_rt -Wl,-rpath,"$MKLDIR" \
  -L"$ROOT/rslmtoasa_amd" -lrsrec -Wl,-rpath,'$ORIGIN/../../rslmtoasa_amd' -Wl,-rpath,/opt/rocm/lib \
  -o "$ROOT/oracle/_ref/rslmto_dropin.x"
echo "built $ROOT/oracle/_ref/rslmto_dropin.x (zero-edit drop-in: the reference's own main program)"
