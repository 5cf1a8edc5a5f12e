!------------------------------------------------------------------------------
! lattice_mod -- SHADOW of the reference's module of the same name (source/lattice.f90), for the zero-edit drop-in build.
!
! The reference's lattice.f90 is compiled unchanged but under another module name (-Dlattice_mod=lattice_ref_mod; its sources are
! compiled with -cpp already), the GPU type of fortran/ extends the reference type from there, and THIS module hands that extended type
! out under the reference's names: every `use lattice_mod` in the reference -- calculation.f90, self.f90, main.f90 and the modules
! between -- then declares and constructs `type(lattice)` objects that ARE `type(lattice_cells)`, with no line of the reference edited.
! The reference module exports nothing but the type and its generic constructor (`private` + `type, public`), and a rename on
! use association carries both.  Recipe: fortran/build_dropin.sh; INTEGRATION.md section 2.
!------------------------------------------------------------------------------
module lattice_mod
   use lattice_cells_mod, only: lattice => lattice_cells
   implicit none
   private
   public :: lattice
end module lattice_mod
