!------------------------------------------------------------------------------
! kubo_gpu_driver -- the reference's conductivity post-processing with the GPU recursion type dropped in.
!
! Mirrors calculation.f90:960-1078 (post_processing_conductivity) with ONE change: the recursion object is a
! `type(recursion_gpu)`, whose compute_moments_stochastic (recursion.f90:979) runs on the GPU (rsrec_kubo_moments).
! Everything else -- setup_kubo_operators, conductivity%calculate_gamma_nm / calculate_conductivity_tensor, the output
! files -- is the reference's compiled code.  End-to-end drop-in test of SURVEY 8 a11 / f4: <label>_cond.out is compared
! with the reference's committed tests/postproc/references/Example_exchange_conductivity_fccPt*/ref.json.
! Run inside a scratch copy of a conductivity case directory; reads input.nml.
!------------------------------------------------------------------------------
program kubo_gpu_driver
   use mpi_mod
   use control_mod
   use lattice_mod
   use charge_mod
   use mix_mod
   use energy_mod
   use hamiltonian_mod
   use recursion_gpu_mod
   use density_of_states_mod
   use green_mod
   use bands_mod
   use self_mod
   use conductivity_mod
   use timer_mod, only: g_timer, timer
   implicit none

   type(control), target :: control_obj
   type(lattice), target :: lattice_obj
   type(energy), target :: energy_obj
   type(self), target :: self_obj
   type(charge), target :: charge_obj
   type(hamiltonian), target :: hamiltonian_obj
   type(recursion_gpu), target :: recursion_obj
   type(green), target :: green_obj
   type(dos), target :: dos_obj
   type(bands), target :: bands_obj
   type(mix), target :: mix_obj
   type(conductivity), target :: conductivity_obj
   integer :: i, elen, estat
   integer(8) :: t0, t1, rate
   character(len=8) :: envv

   rank = 0
   numprocs = 1
   g_timer = timer()
   control_obj = control('input.nml')
   lattice_obj = lattice(control_obj)
   if (control_obj%calctype /= 'B') stop 'kubo_gpu_driver: bulk cases only'
   call lattice_obj%build_data()
   call lattice_obj%bravais()
   call lattice_obj%structb(.true.)
   call lattice_obj%atomlist()
   call get_mpi_variables(rank, lattice_obj%ntype)
   charge_obj = charge(lattice_obj)
   call charge_obj%bulkmat()
   mix_obj = mix(lattice_obj, charge_obj)
   energy_obj = energy(lattice_obj)
   call energy_obj%e_mesh()
   hamiltonian_obj = hamiltonian(charge_obj)
   do i = 1, lattice_obj%nrec
      call lattice_obj%symbolic_atoms(i)%build_pot()
   end do
   if (control_obj%nsp == 2 .or. control_obj%nsp == 4) call hamiltonian_obj%build_lsham
   call hamiltonian_obj%build_bulkham()
   recursion_obj = recursion_gpu(hamiltonian_obj, energy_obj)     ! <-- the one-line change
   ! RSREC_ORBITAL=1: the orbital-moment workflow instead (calculation.f90:1256 after the same set-up): unit 50 is its output
   call get_environment_variable('RSREC_ORBITAL', envv, elen, estat)
   if (estat == 0 .and. elen > 0) then
      call system_clock(t0, rate)
      call recursion_obj%chebyshev_orbital_mod()
      call system_clock(t1)
      flush (50)
      write (*, '(a,f12.6,a)') 'kubo_gpu_driver: chebyshev_orbital_mod wall time ', real(t1 - t0)/real(rate), ' s'
      call g_timer%print_report()
      call rsrec_gpu_shutdown()
      stop
   end if
   call system_clock(t0, rate)
   call recursion_obj%compute_moments_stochastic()
   call system_clock(t1)
   write (*, '(a,f12.6,a)') 'kubo_gpu_driver: compute_moments_stochastic wall time ', real(t1 - t0)/real(rate), ' s'
   write (*, '(a,es14.6,a,es14.6)') 'kubo_gpu_driver: max |mu_nm| = ', maxval(abs(recursion_obj%mu_nm_stochastic)), '  max |mu(:,:,1,1)| = ', &
      maxval(abs(recursion_obj%mu_nm_stochastic(:, :, 1, 1, :)))
   dos_obj = dos(recursion_obj, energy_obj)
   green_obj = green(dos_obj)
   bands_obj = bands(green_obj)
   self_obj = self(bands_obj, mix_obj)
   conductivity_obj = conductivity(self_obj)
   call conductivity_obj%calculate_gamma_nm()
   call conductivity_obj%calculate_conductivity_tensor()
   call rsrec_gpu_shutdown()
end program kubo_gpu_driver
