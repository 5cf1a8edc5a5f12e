!------------------------------------------------------------------------------
! scf_gpu_driver -- the reference's SCF workflow with the GPU recursion type dropped in.
!
! Mirrors the call sequence of calculation.f90 (pre_processing_bravais :550-623, buildsurf :475-543,
! newclubulk :320-390, newclusurf :397-468) with ONE change: the recursion object is a
! `type(recursion_gpu)` (fortran/recursion_gpu.f90).  All other objects are the reference's own compiled
! modules (oracle/_ref/librslmto_ref.a); self%run() is the reference's SCF loop and calls
! recursion%recur_b / chebyshev_recur / recur / zsqr through its class(recursion) pointer (self.f90:799-806, :829),
! which now dispatch to the HIP kernels.  Used as the end-to-end drop-in test: the outputs (<label>_out.nml,
! totaldos.out) are compared with the reference's committed tests/scf/references/*/ref.json values.
! Run inside a scratch copy of a case directory; reads input.nml.
!
! Environment (tests/test_fortran_dropin.py):
!   RSREC_HOST_LDOS=1   bands_gpu%device_ldos = F: calculate_fermi sums a downloaded g0 on the host (the inherited routine)
!   RSREC_HOST_HAM=1    hamiltonian_gpu%device_assembly = F: build_bulkham / build_locham are the reference's host routines
!   RSREC_DEFER_G0=1    green_gpu%defer_g0 = T: g0 is produced only when a routine reads it
!   RSREC_RANK, RSREC_NRANKS, RSREC_COMM_FILE   a run of several processes WITHOUT MPI: this process is rank RSREC_RANK of RSREC_NRANKS
!                       (mpi_mod's rank / numprocs, so get_mpi_variables deals the sites), one GPU each, and the densities of states
!                       are summed over the library's own RCCL communicator whose id travels through the file (rsrec_comm_init_file)
!   RSREC_LDOS_ONLY=1   no SCF loop: one recursion + the density-of-states stage (the call sequence of self.f90:769-806, :821-833 and
!                       calculation.f90:700-712 up to calculate_fermi), then the timer report; with RSREC_DEFER_G0=1 no g0 exists at all
!------------------------------------------------------------------------------
program scf_gpu_driver
   use mpi_mod
   use control_mod
   use lattice_mod
   use lattice_cells_mod
   use charge_mod
   use mix_mod
   use energy_mod
   use hamiltonian_mod
   use recursion_gpu_mod
   use hamiltonian_gpu_mod
   use density_of_states_mod
   use green_mod
   use green_gpu_mod
   use bands_mod
   use bands_gpu_mod
   use self_mod
   use calculation_mod
   use symbolic_atom_mod, only: save_state
   use timer_mod, only: g_timer, timer
   use rsrec_binding
   use, intrinsic :: iso_c_binding
   implicit none

   type(calculation) :: calc_obj
   type(control), target :: control_obj
   type(lattice_cells), target :: lattice_obj   ! <-- third drop-in: O(N) neighbour search behind structb / newclu (CPU)
   type(energy), target :: energy_obj
   type(self), target :: self_obj
   type(charge), target :: charge_obj
   type(hamiltonian_gpu), target :: hamiltonian_obj   ! <-- fourth drop-in: ee / eeo / hall / hallo assembled on the GPU (build_bulkham, build_locham)
   type(recursion_gpu), target :: recursion_obj
   type(green_gpu), target :: green_obj      ! <-- second drop-in: the Green function of the block recursion on the GPU
   type(dos), target :: dos_obj
   type(bands_gpu), target :: bands_obj      ! <-- the density-of-states reduction of calculate_fermi on the GPU, g0 on demand
   type(mix), target :: mix_obj
   character(len=32) :: pre
   character(len=8) :: envv
   character(len=512) :: comm_file
   integer :: ia, elen, estat, crc
   logical :: ldos_only
   real(c_double) :: tinfo(11)

   rank = 0
   numprocs = 1
   call get_environment_variable('RSREC_NRANKS', envv, elen, estat)
   if (estat == 0 .and. elen > 0) read (envv, *) numprocs
   call get_environment_variable('RSREC_RANK', envv, elen, estat)
   if (estat == 0 .and. elen > 0) read (envv, *) rank
   g_timer = timer()
   call g_timer%start('Calculation')

   calc_obj = calculation('input.nml')
   pre = trim(calc_obj%pre_processing)
   control_obj = control('input.nml')
   lattice_obj%lattice = lattice(control_obj)
   call g_timer%start('pre-processing')
   call lattice_obj%build_data()
   call lattice_obj%bravais()
   select case (trim(pre))
   case ('bravais')
      call lattice_obj%structb(.true.)
   case ('buildsurf')
      call lattice_obj%build_surf_full()
      call lattice_obj%structb(.true.)
   case ('newclubulk')
      call lattice_obj%newclu()
      call lattice_obj%structb(.true.)
   case ('newclusurf')
      call lattice_obj%build_surf_full()
      call lattice_obj%newclu()
      call lattice_obj%structb(.true.)
   case default
      stop 'scf_gpu_driver: unsupported pre_processing'
   end select
   call lattice_obj%atomlist()
   call get_mpi_variables(rank, lattice_obj%nrec)
   charge_obj = charge(lattice_obj%lattice)      ! (the constructors take a non-polymorphic type(lattice) dummy)
   select case (trim(pre))
   case ('bravais')
      call charge_obj%bulkmat()
   case ('buildsurf')
      call charge_obj%build_alelay
      call charge_obj%surfmat
   case default
      call charge_obj%impmad()
      call charge_obj%get_charge_transf
   end select
   call g_timer%stop('pre-processing')

   mix_obj = mix(lattice_obj%lattice, charge_obj)
   energy_obj = energy(lattice_obj%lattice)
   hamiltonian_obj%hamiltonian = hamiltonian(charge_obj)
   call get_environment_variable('RSREC_HOST_HAM', envv, elen, estat)
   if (estat == 0 .and. elen > 0) hamiltonian_obj%device_assembly = .false.
   recursion_obj = recursion_gpu(hamiltonian_obj, energy_obj)     ! <-- the one-line change
   dos_obj = dos(recursion_obj, energy_obj)
   green_obj = green_gpu(dos_obj)
   ! bands' constructor takes a non-polymorphic `type(green)` dummy (bands.f90:121); in the reference itself that dummy
   ! becomes `class(green)` (INTEGRATION.md).  With the reference's object code as it is, the parent component is passed
   ! and the class pointer re-pointed at the whole object, which is what the polymorphic dummy would have done.
   bands_obj = bands_gpu(green_obj)
   call get_environment_variable('RSREC_HOST_LDOS', envv, elen, estat)
   if (estat == 0 .and. elen > 0) bands_obj%device_ldos = .false.
   call get_environment_variable('RSREC_DEFER_G0', envv, elen, estat)
   if (estat == 0 .and. elen > 0) green_obj%defer_g0 = .true.
   call get_environment_variable('RSREC_LDOS_ONLY', envv, elen, estat)
   ldos_only = estat == 0 .and. elen > 0
   call get_environment_variable('RSREC_COMM_FILE', comm_file, elen, estat)
   if (estat == 0 .and. elen > 0) then
      crc = rsrec_comm_init_file(rsrec_gpu_context(), int(rank, c_int), int(numprocs, c_int), trim(comm_file)//c_null_char, 120.0_c_double)
      if (crc /= 0) stop 'scf_gpu_driver: rsrec_comm_init_file failed'
      write (*, '(a,i0,a,i0)') 'library communicator: rank ', rank, ' of ', numprocs
   end if
   ! self's constructor has the same non-polymorphic dummy (self.f90:262): parent component in, class pointer re-pointed
   self_obj = self(bands_obj%bands, mix_obj)
   self_obj%bands => bands_obj
   if (ldos_only) then
      ! one recursion and the density-of-states stage, no atomic-sphere step (run_recursion self.f90:769-806; run_dos :821-833)
      call g_timer%start('ldos-only')
      select case (control_obj%calctype)
      case ('B')
         do ia = 1, lattice_obj%nrec
            call lattice_obj%symbolic_atoms(ia)%build_pot()
         end do
      case default
         do ia = 1, lattice_obj%ntype
            call lattice_obj%symbolic_atoms(ia)%build_pot()
         end do
      end select
      if (control_obj%nsp == 2 .or. control_obj%nsp == 4) call hamiltonian_obj%build_lsham
      call hamiltonian_obj%build_bulkham()
      if (control_obj%calctype == 'I') call hamiltonian_obj%build_locham()
      call recursion_obj%recur_b()
      call energy_obj%e_mesh()
      call recursion_obj%zsqr()
      call green_obj%block_green()
      call bands_obj%calculate_fermi()
      call g_timer%stop('ldos-only')
      write (*, '(a,i0,a,l1)') 'ldos-only: device_ldos_calls=', bands_obj%n_device_ldos, ' g0_pending=', green_obj%g0_stale
      write (*, '(a,i0)') 'device_assemblies=', hamiltonian_obj%n_device_assemblies
   else
      call g_timer%start('self-consistency')
      call self_obj%run()
      call g_timer%stop('self-consistency')
      write (*, '(a,i0)') 'device_assemblies=', hamiltonian_obj%n_device_assemblies
      call save_state(lattice_obj%symbolic_atoms)
   end if
   call g_timer%stop('Calculation')
   call g_timer%print_report()
   crc = rsrec_get_timing(rsrec_gpu_context(), tinfo, 11_c_int)
   write (*, '(a,i0)') 'operator_arrays_from_device=', nint(tinfo(11))
   call rsrec_gpu_shutdown()
end program scf_gpu_driver
