!------------------------------------------------------------------------------
! RS-LMTO-ASA drop-in: the stage behind the SCALAR recursion (control%recur = 'lanczos') on the GPU.
!------------------------------------------------------------------------------
!
! MODULE: dos_gpu_mod
!
! DESCRIPTION:
!> `type, extends(dos) :: dos_gpu` overrides `density` (density_of_states.f90:248-363): for one site and direction, the
!> Beer-Pettifor band of each of the 18 scalar chains (bpOPT, recursion.f90:3540) and one continued fraction per orbital and
!> energy (bprldos, density_of_states.f90:370-404).  `green%sgreen` (green.f90:628-705) calls it through its
!> `class(dos), pointer`, so with this type behind the pointer the inherited sgreen builds g0 from device results; everything
!> else of `dos` (chebyshev_dos, chebyshev_dos_full, doscheb) is inherited.  The GPU side is `rsrec_scalar_density`
!> (include/rsrec.h): one thread per chain for the band edges, one per (chain, energy) for the fractions.
!>
!> In the zero-edit build (fortran/build_dropin.sh) fortran/shadow/density_of_states_mod.f90 hands this type out as `dos`, and
!> `type(dos) :: dos_obj ; dos_obj = dos(recursion_obj, energy_obj)` in calculation.f90 constructs it.  A host that declares the
!> types itself passes it wherever a `class(dos)` is taken (the `type(dos)` dummy of green's constructor, green.f90:103, is the one
!> non-polymorphic spot).
!------------------------------------------------------------------------------
module dos_gpu_mod
   use, intrinsic :: iso_c_binding
   use density_of_states_mod
   use recursion_mod
   use energy_mod
   use precision_mod, only: rp
   use logger_mod, only: g_logger
   use timer_mod, only: g_timer
   use rsrec_binding
   use rsrec_context_mod, only: rsrec_gpu_context
   implicit none

   private

   type, public, extends(dos) :: dos_gpu
   contains
      procedure :: density => gpu_density
   end type dos_gpu

   interface dos_gpu
      procedure :: gpu_constructor
   end interface dos_gpu

contains

   !> Same construction as density_of_states.f90:67-79.
   function gpu_constructor(recursion_obj, energy_obj) result(obj)
      type(dos_gpu) :: obj
      class(recursion), target, intent(in) :: recursion_obj
      class(energy), target, intent(in) :: energy_obj

      obj%recursion => recursion_obj
      obj%en => energy_obj
      obj%symbolic_atom => recursion_obj%hamiltonian%charge%lattice%symbolic_atoms
      obj%lattice => recursion_obj%lattice
      obj%control => recursion_obj%lattice%control
      call obj%restore_to_default()
   end function gpu_constructor

   !> Replaces density_of_states.f90:248-363 (same interface).  The trace lines the reference writes to unit 300 (:359-361) are kept.
   subroutine gpu_density(this, tdens, ia, mdir)
      use mpi_mod, only: l2g_map
      class(dos_gpu) :: this
      integer, intent(in) :: mdir
      integer, intent(in) :: ia
      real(rp), dimension(18, this%en%channels_ldos + 10), intent(out) :: tdens
      !
      integer :: npts, llmax, ia_glob, eidx
      integer(c_int) :: rc
      type(c_ptr) :: handle
      real(rp), allocatable, target :: a(:, :), b2(:, :), ene(:), dw(:), cs(:), td(:, :)

      npts = this%en%channels_ldos + 10
      llmax = size(this%recursion%a, 1)
      ia_glob = l2g_map(ia)                                   ! the atom whose potential parameters the reference reads (:252, :329-330)
      allocate (a(llmax, 18), b2(llmax, 18), ene(npts), dw(18), cs(18), td(18, npts))
      a = this%recursion%a(:, :, ia, mdir)
      b2 = this%recursion%b2(:, :, ia, mdir)
      ene = this%en%ene(1:npts)
      dw = this%symbolic_atom(ia_glob)%potential%dw_l(1:18)
      cs = this%symbolic_atom(ia_glob)%potential%cshi(1:18)
      handle = rsrec_gpu_context()
      call g_timer%start('density-gpu')
      rc = rsrec_scalar_density(handle, 1_c_int, 1_c_int, int(llmax, c_int), int(this%control%lld, c_int), c_loc(a), c_loc(b2), int(npts, c_int), &
                                c_loc(ene), c_loc(dw), c_loc(cs), c_loc(td))
      call g_timer%stop('density-gpu')
      if (rc /= 0) call g_logger%fatal('rsrec_scalar_density: '//rsrec_error_string(handle), __FILE__, __LINE__)
      tdens = td
      do eidx = 1, npts
         write (300, *) this%en%ene(eidx), sum(tdens(1:9, eidx)), sum(tdens(10:18, eidx))
      end do
   end subroutine gpu_density
end module dos_gpu_mod
