! TEST INFRASTRUCTURE ONLY (oracle) -- compiled only in the build container.  The binary (oracle/_ref/ref_kernel.x) travels to the
! GPU box as the CPU leg of bench.py (cpu_baseline.kind = "reference") and as a checker of the -m gpu tests; nothing under
! rslmtoasa_amd/ uses it.
!
! Reference-kernel driver: feeds a recursion problem (lattice tables + Hamiltonian blocks read from
! `kernel_in.bin`, written by oracle/fixture_io.py) straight into the *compiled reference*
! `recursion_mod` (oracle/_ref/librslmto_ref.a) and writes what it produces to `kernel_out.bin`.
! It bypasses the reference's O(kk^2) cluster builder (lattice.f90:3035) so that the reference's
! own hot-path code (recursion.f90:1807 recur_b, :3057 chebyshev_recur, :3485 recur) can be run --
! and timed with its own g_timer regions (recursion.f90:1902-1970) -- on synthetic periodic
! supercells of any size.  All recursion arithmetic executed here is the reference's.
program ref_kernel
   use mpi_mod
   use control_mod
   use lattice_mod
   use charge_mod
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
   integer :: u, magic, version, kk, nncols, nmax, ntype, nrec, lld, nsp, hoh_i, kind_rec, nslots, llmax, npairs, nchains
   real(rp) :: emin, emax
   integer :: t0, t1, rate

   rank = 0
   numprocs = 1
   g_timer = timer()

   open (newunit=u, file='kernel_in.bin', access='stream', form='unformatted', status='old')
   read (u) magic, version
   if (magic /= int(z'52534658')) stop 'ref_kernel: bad magic'
   read (u) kk, nncols, nmax, ntype, nrec, lld, nsp, hoh_i, kind_rec, nslots, llmax
   read (u) emin, emax

   control_obj%lld = lld
   control_obj%llsp = lld
   control_obj%nsp = nsp
   control_obj%calctype = 'B'
   if (nmax > 0) control_obj%calctype = 'I'
   select case (kind_rec)
   case (0); control_obj%recur = 'block'
   case (1); control_obj%recur = 'chebyshev'
   case (2); control_obj%recur = 'lanczos'
   case (3); control_obj%recur = 'block'        ! recur_b_ij: four chains per atom pair (recursion.f90:1655)
   case (4); control_obj%recur = 'chebyshev'    ! chebyshev_recur_ij (recursion.f90:2376)
   end select

   lattice_obj%control => control_obj
   lattice_obj%kk = kk
   lattice_obj%nmax = nmax
   lattice_obj%ntype = ntype
   lattice_obj%nrec = nrec
   lattice_obj%njij = 0
   lattice_obj%njijk = 0
   allocate (lattice_obj%iz(kk), lattice_obj%nn(kk, nncols), lattice_obj%irec(nrec))
   read (u) lattice_obj%iz
   read (u) lattice_obj%nn
   read (u) lattice_obj%irec
   npairs = 0
   if (kind_rec >= 3) then
      ! for the pair variants `irec` carries the pairs: (i_1, j_1, i_2, j_2, ...)
      npairs = nrec/2
      lattice_obj%njij = npairs
      allocate (lattice_obj%ijpair(npairs, 2))
      lattice_obj%ijpair(:, 1) = lattice_obj%irec(1:nrec:2)
      lattice_obj%ijpair(:, 2) = lattice_obj%irec(2:nrec:2)
   end if

   charge_obj%lattice => lattice_obj
   hamiltonian_obj%charge => charge_obj
   hamiltonian_obj%lattice => lattice_obj
   hamiltonian_obj%control => control_obj
   hamiltonian_obj%hoh = (hoh_i /= 0)
   hamiltonian_obj%local_axis = .false.
   allocate (hamiltonian_obj%ee(18, 18, nslots, ntype), hamiltonian_obj%lsham(18, 18, ntype))
   allocate (hamiltonian_obj%eeo(18, 18, nslots, ntype), hamiltonian_obj%enim(18, 18, ntype))
   allocate (hamiltonian_obj%hall(18, 18, nslots, max(nmax, 0)), hamiltonian_obj%hallo(18, 18, nslots, max(nmax, 0)))
   read (u) hamiltonian_obj%ee
   read (u) hamiltonian_obj%lsham
   read (u) hamiltonian_obj%eeo
   read (u) hamiltonian_obj%enim
   if (nmax > 0) then
      read (u) hamiltonian_obj%hall
      read (u) hamiltonian_obj%hallo
   end if
   close (u)

   energy_obj%energy_min = emin
   energy_obj%energy_max = emax

   if (kind_rec >= 3) then
      call get_mpi_variables(rank, npairs)
   else
      call get_mpi_variables(rank, nrec)
   end if
   recursion_obj = recursion(hamiltonian_obj, energy_obj)

   call system_clock(t0, rate)
   call g_timer%start('recursion')
   select case (kind_rec)
   case (0)
      call recursion_obj%recur_b()
   case (1)
      call recursion_obj%chebyshev_recur()
   case (2)
      call recursion_obj%recur()
   case (3)
      call recursion_obj%recur_b_ij()
   case (4)
      call recursion_obj%chebyshev_recur_ij()
   end select
   call g_timer%stop('recursion')
   call system_clock(t1)
   write (*, '(a,f12.6,a)') 'ref_kernel: recursion wall time ', real(t1 - t0, rp)/real(rate, rp), ' s'
   call g_timer%print_report()

   open (newunit=u, file='kernel_out.bin', access='stream', form='unformatted', status='replace')
   write (u) int(z'52534658'), 1, kind_rec
   select case (kind_rec)
   case (0, 3)
      write (u) recursion_obj%a_b
      write (u) recursion_obj%b2_b
   case (1, 4)
      write (u) recursion_obj%mu_n
   case (2)
      write (u) recursion_obj%a(:, :, :, 1)
      write (u) recursion_obj%b2(:, :, :, 1)
   end select
   close (u)
end program ref_kernel
