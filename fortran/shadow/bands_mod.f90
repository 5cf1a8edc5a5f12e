!------------------------------------------------------------------------------
! bands_mod -- SHADOW of the reference's module of the same name (source/bands.f90), for the zero-edit drop-in build.
!
! The reference's bands.f90 is compiled unchanged but under another module name (-Dbands_mod=bands_ref_mod; its sources are
! compiled with -cpp already), the GPU type of fortran/ extends the reference type from there, and THIS module hands that extended type
! out under the reference's names: every `use bands_mod` in the reference -- calculation.f90, self.f90, main.f90 and the modules
! between -- then declares and constructs `type(bands)` objects that ARE `type(bands_gpu)`, with no line of the reference edited.
! The reference module exports nothing but the type and its generic constructor (`private` + `type, public`), and a rename on
! use association carries both.  Recipe: fortran/build_dropin.sh; INTEGRATION.md section 2.
!------------------------------------------------------------------------------
module bands_mod
   use bands_gpu_mod, only: bands => bands_gpu
   implicit none
   private
   public :: bands
end module bands_mod
