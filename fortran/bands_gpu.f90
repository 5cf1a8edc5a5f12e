!------------------------------------------------------------------------------
! RS-LMTO-ASA drop-in, third stage: the density-of-states reduction of `bands%calculate_fermi` on the GPU.
!------------------------------------------------------------------------------
!
! MODULE: bands_gpu_mod
!
! DESCRIPTION:
!> `type, extends(bands) :: bands_gpu` overrides `calculate_fermi` (bands.f90:227-346).  The reference forms
!>    dtot(i), dosia(site, i), dosial(site, 1:18, i) = -Im g0(j, j, i, site) / pi        (bands.f90:258-268)
!> from the full `g0(18,18,nE,site)` -- 13 MB per site that a GPU Green function first has to send over PCIe.  With a
!> `recursion_gpu` behind the class pointer the block coefficients of the rank's sites are still on the device after `recur_b`
!> (`rsrec_gpu_block_resident`), and `rsrec_block_ldos` (include/rsrec.h) runs zsqr -> get_terminf -> bgreen -> this reduction
!> there: 18 doubles per site and energy come back instead of 648, summed in the reference's loop order.  Everything after the
!> reduction -- the MPI_ALLREDUCE of the zero-padded arrays (:271-274), the three output files (:279-324), the Fermi level
!> (:326-343) -- is the reference's, restated here because `calculate_fermi` is one routine.
!>
!> `g0` on demand: the other consumers of `g0` in `bands` (magnetic / orbital moments, band moments, projected DOS) are inherited;
!> the overrides below only make sure `g0` exists first (`green_gpu%fetch_g0`, a no-op unless `green_gpu%defer_g0` postponed it).
!> An SCF iteration reads `g0` in `calculate_magnetic_moments`, so it is produced once per iteration either way; a flow that stops
!> at the densities of states (calculation.f90:700-712 `block_green` + `calculate_fermi`) never produces it.
!>
!> Falls back to the inherited routine whenever the device does not hold this call's coefficients (Chebyshev / scalar recursion,
!> local-axis runs, a `green` that is not `green_gpu`).
!------------------------------------------------------------------------------
module bands_gpu_mod
   use, intrinsic :: iso_c_binding
   use mpi_mod
   use bands_mod
   use green_mod
   use green_gpu_mod
   use precision_mod, only: rp
   use logger_mod, only: g_logger
   use timer_mod, only: g_timer
   use string_mod, only: fmt
   use rsrec_binding
   use rsrec_context_mod, only: rsrec_gpu_context, rsrec_env_flag
   use recursion_gpu_mod, only: rsrec_gpu_block_resident
#ifdef USE_MPI
   use mpi
#endif
   implicit none

   private

   type, public, extends(bands) :: bands_gpu
      !> .false.: always the inherited host reduction (a g0 is then needed)
      logical :: device_ldos = .true.
      !> number of calculate_fermi calls served by the device stage (diagnostics / tests)
      integer :: n_device_ldos = 0
   contains
      procedure :: calculate_fermi => gpu_calculate_fermi
      procedure :: calculate_magnetic_moments => gpu_calculate_magnetic_moments
      procedure :: calculate_orbital_moments => gpu_calculate_orbital_moments
      procedure :: calculate_orbital_quadrupoles => gpu_calculate_orbital_quadrupoles
      procedure :: calculate_moments => gpu_calculate_moments
      procedure :: calculate_projected_green => gpu_calculate_projected_green
      procedure :: calculate_projected_dos => gpu_calculate_projected_dos
      procedure :: calculate_orbital_dos => gpu_calculate_orbital_dos
   end type bands_gpu

   interface bands_gpu
      procedure :: gpu_constructor
   end interface bands_gpu

contains

   !> Same construction as bands.f90:119-133; the dummy is polymorphic, which is what the reference's constructor needs to become
   !> (INTEGRATION.md) for `green_gpu` to survive it.
   function gpu_constructor(green_obj) result(obj)
      type(bands_gpu) :: obj
      class(green), target, intent(in) :: green_obj

      obj%green => green_obj
      obj%lattice => green_obj%dos%recursion%lattice
      obj%symbolic_atom => green_obj%dos%recursion%hamiltonian%charge%lattice%symbolic_atoms
      obj%dos => green_obj%dos
      obj%en => green_obj%dos%en
      obj%control => green_obj%dos%recursion%lattice%control
      obj%recursion => green_obj%dos%recursion
      call obj%restore_to_default()
      if (rsrec_env_flag('RSREC_HOST_LDOS')) obj%device_ldos = .false.   ! (hosts that cannot reach the member: fortran/shadow/)
   end function gpu_constructor

   !> `g0` must exist before an inherited routine reads it
   subroutine ensure_g0(this)
      class(bands_gpu), intent(inout) :: this
      select type (g => this%green)
      class is (green_gpu)
         call g%fetch_g0()
      end select
   end subroutine ensure_g0

   !> .true. if the device holds the block coefficients this call's densities of states are made of
   function device_stage_usable(this) result(ok)
      class(bands_gpu), intent(in) :: this
      logical :: ok
      ok = this%device_ldos .and. trim(this%control%recur) == 'block' .and. end_atom >= start_atom
      if (ok) ok = rsrec_gpu_block_resident() == end_atom - start_atom + 1
      if (ok) then
         select type (g => this%green)
         class is (green_gpu)
            ok = .true.
         class default
            ok = .false.
         end select
      end if
   end function device_stage_usable

   !---------------------------------------------------------------------------
   !> Total / site / orbital densities of states and the Fermi level (replaces bands.f90:227-346)
   !---------------------------------------------------------------------------
   subroutine gpu_calculate_fermi(this)
      class(bands_gpu) :: this
      integer :: nv, nrec, ia, ik1, ik1_mag, ifail
      integer(c_int) :: rc, sym_i, crank, cranks
      type(c_ptr) :: handle
      real(rp) :: e1, e1_mag, ef_mag
      real(rp), allocatable, target :: ene(:), dtot(:), dosia(:, :), dosial(:, :, :)

      if (.not. device_stage_usable(this)) then
         call ensure_g0(this)
         call this%bands%calculate_fermi()
         return
      end if

      nv = this%en%channels_ldos + 10
      nrec = this%lattice%nrec
      allocate (ene(nv), dtot(nv), dosia(nrec, nv), dosial(nrec, 18, nv))
      ene = this%en%ene(1:nv)
      sym_i = 0
      if (this%control%sym_term) sym_i = 1
      handle = rsrec_gpu_context()
      ! zero-padded images over all nrec sites, this rank's sites start_atom .. end_atom filled (what bands.f90:258-268 leaves
      ! in dtot / dosia / dosial before the all-reduce); sums in the reference's loop order
      call g_timer%start('ldos-gpu')
      rc = rsrec_block_ldos(handle, int(nv, c_int), c_loc(ene), 0.0_c_double, 0.0_c_double, sym_i, int(start_atom - 1, c_int), int(nrec, c_int), &
                            c_loc(dtot), c_loc(dosia), c_loc(dosial), c_null_ptr, c_null_ptr)
      call g_timer%stop('ldos-gpu')
      if (rc /= 0) call g_logger%fatal('rsrec_block_ldos: '//rsrec_error_string(handle), __FILE__, __LINE__)
      this%n_device_ldos = this%n_device_ldos + 1
      this%dtot(1:nv) = dtot

      this%qqv = real(sum(this%symbolic_atom(1:this%lattice%nbulk_bulk)%element%valence))      ! bands.f90:251
      if (rank == 0) call g_logger%info('Valence is:'//fmt('f16.6', this%qqv), __FILE__, __LINE__)
      ! the all-reduce of bands.f90:271-274: over the library's own communicator if the host set one up (rsrec_comm_init[_file]: RCCL
      ! over xGMI, no MPI needed), else the reference's MPI calls
      rc = rsrec_comm_size(handle, crank, cranks)
      if (cranks > 1) then
         dtot = this%dtot(1:nv)
         rc = rsrec_allreduce_sum(handle, c_loc(dtot), int(nv, c_size_t))
         if (rc == 0) rc = rsrec_allreduce_sum(handle, c_loc(dosia), int(size(dosia), c_size_t))
         if (rc == 0) rc = rsrec_allreduce_sum(handle, c_loc(dosial), int(size(dosial), c_size_t))
         if (rc /= 0) call g_logger%fatal('rsrec_allreduce_sum: '//rsrec_error_string(handle), __FILE__, __LINE__)
         this%dtot(1:nv) = dtot
      else
#ifdef USE_MPI
      call MPI_ALLREDUCE(MPI_IN_PLACE, this%dtot, nv, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)       ! bands.f90:271-274
      call MPI_ALLREDUCE(MPI_IN_PLACE, dosia, size(dosia), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
      call MPI_ALLREDUCE(MPI_IN_PLACE, dosial, size(dosial), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr)
#endif
      end if
      ! output files of bands.f90:279-324 (same names, units, formats; every rank replaces totaldos.out, rank 0 fills the files)
      call write_columns(125, 'totaldos.out', nv, this%en%ene(1:nv) - this%en%fermi, reshape(this%dtot(1:nv), [1, nv]), rank == 0, .true.)
      do ia = 1, nrec
         call write_columns(250 + ia, trim(this%symbolic_atom(this%lattice%nbulk + ia)%element%symbol)//'_dos.out', nv, &
                            this%en%ene(1:nv) - this%en%fermi, reshape(dosia(ia, :), [1, nv]), rank == 0, rank == 0)
         call write_columns(450 + ia, trim(this%symbolic_atom(this%lattice%nbulk + ia)%element%symbol)//'_orbital_dos.out', nv, &
                            this%en%ene(1:nv) - this%en%fermi, dosial(ia, :, :), rank == 0, rank == 0)
      end do

      ! Fermi level (bands.f90:326-343)
      ik1 = this%en%ik1
      ik1_mag = 0
      ef_mag = this%en%fermi
      this%en%chebfermi = this%en%fermi
      if (.not. (this%en%fix_fermi) .and. this%control%calctype == 'B') then
         e1_mag = ef_mag
         call this%fermi(ef_mag, this%en%edel, ik1_mag, this%en%energy_min, nv, this%dtot, ifail, this%qqv, e1_mag)
         e1 = this%en%fermi
         call this%fermi(this%en%fermi, this%en%edel, ik1, this%en%energy_min, nv, this%dtot, ifail, this%qqv, e1_mag)
         this%nv1 = ik1
         this%e1 = e1_mag
         if (rank == 0) call g_logger%info('Free Fermi energy:'//fmt('f10.6', this%en%fermi), __FILE__, __LINE__)
      else if (this%en%fix_fermi) then
         ik1 = nint((this%en%fermi - this%en%energy_min)/this%en%edel)
         e1 = this%en%energy_min + (ik1 - 1)*this%en%edel
         this%nv1 = ik1
         this%e1 = e1
         if (rank == 0) call g_logger%info('Fixed Fermi energy:'//fmt('f10.6', this%en%fermi), __FILE__, __LINE__)
      end if
   end subroutine gpu_calculate_fermi

   !> One of the reference's DOS files: column 1 = x(i), then y(:, i), format (<1 + rows>f16.5), opened with status 'replace'.
   subroutine write_columns(unitnum, fname, n, x, y, fill, opened)
      integer, intent(in) :: unitnum, n
      character(len=*), intent(in) :: fname
      real(rp), intent(in) :: x(n), y(:, :)
      logical, intent(in) :: fill, opened
      integer :: i
      character(len=16) :: form

      if (.not. opened) return
      write (form, '(a,i0,a)') '(', size(y, 1) + 1, 'f16.5)'
      open (unit=unitnum, file=fname, status='replace', action='write')
      if (fill) then
         do i = 1, n
            write (unitnum, form) x(i), y(:, i)
         end do
         rewind (unitnum)
      end if
      close (unitnum)
   end subroutine write_columns

   ! ---- inherited consumers of g0: make sure it exists, then the reference's routine ------------------------------------
   subroutine gpu_calculate_magnetic_moments(this)
      class(bands_gpu) :: this
      call ensure_g0(this)
      call this%bands%calculate_magnetic_moments()
   end subroutine gpu_calculate_magnetic_moments

   subroutine gpu_calculate_orbital_moments(this)
      class(bands_gpu) :: this
      call ensure_g0(this)
      call this%bands%calculate_orbital_moments()
   end subroutine gpu_calculate_orbital_moments

   subroutine gpu_calculate_orbital_quadrupoles(this)
      class(bands_gpu) :: this
      call ensure_g0(this)
      call this%bands%calculate_orbital_quadrupoles()
   end subroutine gpu_calculate_orbital_quadrupoles

   subroutine gpu_calculate_moments(this)
      class(bands_gpu) :: this
      call ensure_g0(this)
      call this%bands%calculate_moments()
   end subroutine gpu_calculate_moments

   subroutine gpu_calculate_projected_green(this)
      class(bands_gpu) :: this
      call ensure_g0(this)
      call this%bands%calculate_projected_green()
   end subroutine gpu_calculate_projected_green

   subroutine gpu_calculate_projected_dos(this)
      class(bands_gpu) :: this
      call ensure_g0(this)
      call this%bands%calculate_projected_dos()
   end subroutine gpu_calculate_projected_dos

   subroutine gpu_calculate_orbital_dos(this)
      class(bands_gpu) :: this
      call ensure_g0(this)
      call this%bands%calculate_orbital_dos()
   end subroutine gpu_calculate_orbital_dos

end module bands_gpu_mod
