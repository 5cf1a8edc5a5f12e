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
   use density_of_states_mod
   use green_mod
   use green_gpu_mod
   use bands_mod
   use self_mod
   use calculation_mod
   use symbolic_atom_mod, only: save_state
   use timer_mod, only: g_timer, timer
   implicit none

   type(calculation) :: calc_obj
   type(control), target :: control_obj
   type(lattice_cells), target :: lattice_obj   ! <-- third drop-in: O(N) neighbour search behind structb / newclu (CPU)
   type(energy), target :: energy_obj
   type(self), target :: self_obj
   type(charge), target :: charge_obj
   type(hamiltonian), target :: hamiltonian_obj
   type(recursion_gpu), target :: recursion_obj
   type(green_gpu), target :: green_obj      ! <-- second drop-in: the Green function of the block recursion on the GPU
   type(dos), target :: dos_obj
   type(bands), target :: bands_obj
   type(mix), target :: mix_obj
   character(len=32) :: pre

   rank = 0
   numprocs = 1
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
   hamiltonian_obj = hamiltonian(charge_obj)
   recursion_obj = recursion_gpu(hamiltonian_obj, energy_obj)     ! <-- the one-line change
   dos_obj = dos(recursion_obj, energy_obj)
   green_obj = green_gpu(dos_obj)
   ! bands' constructor takes a non-polymorphic `type(green)` dummy (bands.f90:121); in the reference itself that dummy
   ! becomes `class(green)` (INTEGRATION.md).  With the reference's object code as it is, the parent component is passed
   ! and the class pointer re-pointed at the whole object, which is what the polymorphic dummy would have done.
   bands_obj = bands(green_obj%green)
   bands_obj%green => green_obj
   self_obj = self(bands_obj, mix_obj)
   call g_timer%start('self-consistency')
   call self_obj%run()
   call g_timer%stop('self-consistency')
   call save_state(lattice_obj%symbolic_atoms)
   call g_timer%stop('Calculation')
   call g_timer%print_report()
   call rsrec_gpu_shutdown()
end program scf_gpu_driver
