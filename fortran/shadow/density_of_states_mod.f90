!------------------------------------------------------------------------------
! density_of_states_mod -- SHADOW of the reference's module of the same name (source/density_of_states.f90), for the zero-edit drop-in build.
!
! The reference's density_of_states.f90 is compiled unchanged but under another module name
! (-Ddensity_of_states_mod=density_of_states_ref_mod), the GPU type of fortran/dos_gpu.f90 extends the reference type from there, and THIS
! module hands that extended type out under the reference's names: `type(dos) :: dos_obj ; dos_obj = dos(recursion_obj, energy_obj)` in
! calculation.f90 and the `type(dos)` dummy of green's constructor (green.f90:103) then are `type(dos_gpu)`, with no line of the reference
! edited.  The reference module exports nothing but the type and its generic constructor, and a rename on use association carries both.
! Recipe: fortran/build_dropin.sh; INTEGRATION.md section 2.
!------------------------------------------------------------------------------
module density_of_states_mod
   use dos_gpu_mod, only: dos => dos_gpu
   implicit none
   private
   public :: dos
end module density_of_states_mod
