! TEST INFRASTRUCTURE ONLY (oracle) -- compiled only in the build container (oracle/build_ref.sh).
!
! Fixture generator for the stochastic Kubo path: links against the compiled reference modules and replays the reference's
! own set-up of post_processing_conductivity (calculation.f90:960-1052) up to and including
! recursion%compute_moments_stochastic (recursion.f90:979-1234), then writes that routine's INPUTS (lattice tables, Hamiltonian
! blocks, the velocity operators setup_kubo_operators :242 left in hamiltonian%v_a / v_b / vo_a / vo_b, seeds, scale / shift)
! and its OUTPUT mu_nm_stochastic at full precision to `kubo.bin`.  All arithmetic is the reference's compiled code.
! Run inside a scratch copy of a conductivity case directory; cond_calctype = 'per_type' (the seeds are deterministic).
program dump_kubo
   use mpi_mod
   use control_mod
   use lattice_mod
   use charge_mod
   use mix_mod
   use energy_mod
   use hamiltonian_mod
   use recursion_mod
   use precision_mod, only: rp
   use timer_mod, only: g_timer, timer
   implicit none

   type(control), target :: control_obj
   type(lattice), target :: lattice_obj
   type(energy), target :: energy_obj
   type(charge), target :: charge_obj
   type(hamiltonian), target :: hamiltonian_obj
   type(recursion), target :: recursion_obj
   type(mix), target :: mix_obj
   integer :: i, k, u, hoh_i, nslots, elen, estat, version
   real(rp), allocatable :: rng(:, :)
   integer(8) :: t0, t1, rate
   real(rp) :: a, b
   character(len=8) :: envv

   rank = 0
   numprocs = 1
   g_timer = timer()

   control_obj = control('input.nml')
   lattice_obj = lattice(control_obj)
   if (control_obj%calctype /= 'B') stop 'dump_kubo: bulk cases only'
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
   recursion_obj = recursion(hamiltonian_obj, energy_obj)

   ! RSREC_DUMP_ORBITAL=1: the other whole-vector routine of this set-up, chebyshev_orbital_mod (recursion.f90:2834-3049, called at
   ! calculation.f90:1256 after exactly the steps above).  Its only outputs are unit 50 (energy, integrated and energy-resolved orbital
   ! moment, 3es16.6) and stdout; `orbital.bin` carries the inputs.
   call get_environment_variable('RSREC_DUMP_ORBITAL', envv, elen, estat)
   if (estat == 0 .and. elen > 0) then
      call system_clock(t0, rate)
      call recursion_obj%chebyshev_orbital_mod()
      call system_clock(t1)
      write (*, '(a,f12.6,a)') 'dump_kubo: chebyshev_orbital_mod wall time ', real(t1 - t0, rp)/real(rate, rp), ' s'
      flush (50)
      a = (energy_obj%energy_max - energy_obj%energy_min)/(2 - 0.3)
      b = (energy_obj%energy_max + energy_obj%energy_min)/2
      nslots = size(hamiltonian_obj%ee, 3)
      hoh_i = 0
      if (hamiltonian_obj%hoh) hoh_i = 1
      open (newunit=u, file='orbital.bin', access='stream', form='unformatted', status='replace')
      write (u) int(z'4f52424d'), 1
      write (u) lattice_obj%kk, size(lattice_obj%nn, 2), lattice_obj%nmax, lattice_obj%ntype, control_obj%lld, control_obj%nsp, hoh_i, nslots, &
         energy_obj%channels_ldos + 10, energy_obj%nv1
      write (u) a, b, lattice_obj%alat, energy_obj%fermi
      write (u) lattice_obj%iz(1:lattice_obj%kk)
      write (u) lattice_obj%nn
      write (u) hamiltonian_obj%ee
      write (u) hamiltonian_obj%lsham
      write (u) hamiltonian_obj%eeo
      write (u) hamiltonian_obj%enim
      write (u) lattice_obj%cr(1:3, 1:lattice_obj%kk)
      write (u) energy_obj%ene(1:energy_obj%channels_ldos + 10)
      close (u)
      stop
   end if

   call system_clock(t0, rate)
   call recursion_obj%compute_moments_stochastic()
   call system_clock(t1)
   write (*, '(a,f12.6,a)') 'dump_kubo: compute_moments_stochastic wall time ', real(t1 - t0, rp)/real(rate, rp), ' s'

   a = (energy_obj%energy_max - energy_obj%energy_min)/(2 - 0.3)
   b = (energy_obj%energy_max + energy_obj%energy_min)/2
   nslots = size(hamiltonian_obj%ee, 3)
   hoh_i = 0
   if (hamiltonian_obj%hoh) hoh_i = 1
   open (newunit=u, file='kubo.bin', access='stream', form='unformatted', status='replace')
   ! version 3 (cond_calctype = 'random_vec', recursion.f90:1101-1140): the random numbers of every vector.  The routine draws them as
   ! `call random_seed()` + kk x `call random_number(rng)` per vector (:1106, :1130-1136); this toolchain's random_seed() without
   ! arguments RESETS the generator to its default state (amdflang / flang runtime -- checked: the same numbers in every process and after
   ! every call), so replaying the same calls here yields exactly the numbers the routine used.  (With a runtime that seeds from the
   ! clock this record would be meaningless; the fixture generator checks that the moments computed from it reproduce mu_nm_stochastic.)
   version = 2
   if (trim(control_obj%cond_calctype) == 'random_vec') then
      version = 3
      allocate (rng(lattice_obj%kk, size(recursion_obj%mu_nm_stochastic, 5)))
      do i = 1, size(rng, 2)
         call random_seed()
         do k = 1, lattice_obj%kk
            call random_number(rng(k, i))
         end do
      end do
   end if
   write (u) int(z'4b55424f'), version
   write (u) lattice_obj%kk, size(lattice_obj%nn, 2), lattice_obj%nmax, lattice_obj%ntype, control_obj%cond_ll, control_obj%nsp, hoh_i, nslots, &
      size(recursion_obj%mu_nm_stochastic, 5)
   write (u) a, b
   write (u) lattice_obj%iz(1:lattice_obj%kk)
   write (u) lattice_obj%nn
   write (u) lattice_obj%atlist(1:lattice_obj%ntype)
   write (u) hamiltonian_obj%ee
   write (u) hamiltonian_obj%lsham
   write (u) hamiltonian_obj%eeo
   write (u) hamiltonian_obj%enim
   write (u) hamiltonian_obj%v_a
   write (u) hamiltonian_obj%v_b
   write (u) hamiltonian_obj%vo_a
   write (u) hamiltonian_obj%vo_b
   write (u) recursion_obj%mu_nm_stochastic
   write (u) lattice_obj%cr(1:3, 1:lattice_obj%kk), lattice_obj%alat          ! version 2: positions (units of alat) and alat
   if (version >= 3) write (u) rng
   close (u)
end program dump_kubo
