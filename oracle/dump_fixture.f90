! TEST INFRASTRUCTURE ONLY (oracle) -- compiled and run only in the build container (the fixtures it writes are committed).
!
! Fixture generator: links against the *compiled reference modules* (oracle/_ref/librslmto_ref.a,
! built by oracle/build_ref.sh from the sources where they lie) and replays the reference's own
! set-up sequence up to the recursion call, then runs the reference recursion driver and writes
! the recursion's INPUTS (lattice tables + Hamiltonian blocks) and OUTPUTS (coefficients/moments)
! at full precision to a stream file `fixture.bin`.
!
! Sequence mirrored (call order only; all arithmetic is the reference's own compiled code):
!   calculation.f90:550-611  (pre_processing_bravais),  :475-543 (buildsurf),
!   calculation.f90:320-390  (newclubulk),              :397-468 (newclusurf)
!   self.f90:769-806         (run_recursion: build_pot / build_lsham / build_bulkham / build_locham,
!                             then recur | recur_b | chebyshev_recur)
! Run it inside a scratch copy of a case directory (input.nml + <label>.nml): it reads input.nml.
program dump_fixture
   use mpi_mod
   use control_mod
   use lattice_mod
   use charge_mod
   use mix_mod
   use energy_mod
   use hamiltonian_mod
   use recursion_mod
   use density_of_states_mod
   use green_mod
   use bands_mod
   use self_mod
   use calculation_mod
   use precision_mod, only: rp
   use timer_mod, only: g_timer, timer
   use logger_mod, only: g_logger
   implicit none

   type(calculation) :: calc_obj
   type(control), target :: control_obj
   type(lattice), target :: lattice_obj
   type(energy), target :: energy_obj
   type(charge), target :: charge_obj
   type(hamiltonian), target :: hamiltonian_obj
   type(recursion), target :: recursion_obj
   type(mix), target :: mix_obj
   type(dos), target :: dos_obj
   type(green), target :: green_obj
   real(rp), allocatable :: a_inf(:, :, :), b_inf(:, :, :), a_inf0(:), b_inf0(:)
   integer :: nw, nen, sym_i, ieta, fpt
   integer(8) :: tc0, tc1, tc2, trate
   integer, parameter :: neta = 5
   complex(rp) :: eta_c
   complex(rp), allocatable :: g_ef(:, :, :)
   real(rp), allocatable :: doso(:, :)
   integer :: ia, u, kind_rec, nslots, hoh_i, nsites, ncheb
   real(rp) :: acheb, bcheb
   character(len=32) :: pre, envbuf

   rank = 0
   numprocs = 1
   g_timer = timer()

   calc_obj = calculation('input.nml')
   pre = trim(calc_obj%pre_processing)

   control_obj = control('input.nml')
   lattice_obj = lattice(control_obj)
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
      stop 'dump_fixture: unsupported pre_processing'
   end select
   call lattice_obj%atomlist()
   call get_mpi_variables(rank, lattice_obj%nrec)

   charge_obj = charge(lattice_obj)
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

   mix_obj = mix(lattice_obj, charge_obj)
   energy_obj = energy(lattice_obj)
   hamiltonian_obj = hamiltonian(charge_obj)
   recursion_obj = recursion(hamiltonian_obj, energy_obj)

   ! ---- self.f90:769-797 (run_recursion, operator set-up part)
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

   ! ---- self.f90:799-806 (the recursion itself = the hot path under test)
   select case (trim(control_obj%recur))
   case ('lanczos')
      kind_rec = 2
      call recursion_obj%recur()
   case ('chebyshev')
      kind_rec = 1
      call recursion_obj%chebyshev_recur()
   case ('block')
      kind_rec = 0
      call recursion_obj%recur_b()
   case default
      stop 'dump_fixture: unsupported recur'
   end select

   nslots = size(hamiltonian_obj%ee, 3)
   hoh_i = 0
   if (hamiltonian_obj%hoh) hoh_i = 1
   nsites = lattice_obj%nrec
   ncheb = 2*control_obj%lld + 2
   acheb = (energy_obj%energy_max - energy_obj%energy_min)/(2 - 0.3)
   bcheb = (energy_obj%energy_max + energy_obj%energy_min)/2

   open (newunit=u, file='fixture.bin', access='stream', form='unformatted', status='replace')
   write (u) int(z'52534658'), 1
   write (u) lattice_obj%kk, size(lattice_obj%nn, 2), lattice_obj%nmax, lattice_obj%ntype, lattice_obj%nrec, &
      control_obj%lld, control_obj%nsp, hoh_i, kind_rec, nslots, size(recursion_obj%a, 1)
   write (u) energy_obj%energy_min, energy_obj%energy_max, acheb, bcheb
   write (u) lattice_obj%iz(1:lattice_obj%kk)
   write (u) lattice_obj%nn
   write (u) lattice_obj%irec(1:lattice_obj%nrec)
   write (u) lattice_obj%cr(1:3, 1:lattice_obj%kk)
   write (u) hamiltonian_obj%ee
   write (u) hamiltonian_obj%lsham
   write (u) hamiltonian_obj%eeo
   write (u) hamiltonian_obj%enim
   if (lattice_obj%nmax > 0) then
      write (u) hamiltonian_obj%hall
      write (u) hamiltonian_obj%hallo
   end if
   select case (kind_rec)
   case (0)
      write (u) recursion_obj%a_b
      write (u) recursion_obj%b2_b
      ! ---- the stage right after the recursion (self.f90:820-831, run_dos): zsqr, then green%block_green; appended so
      !      that the Green-function kernel (SURVEY 8f1) can be pinned: energies, terminator, sqrt(B^2), g0
      dos_obj = dos(recursion_obj, energy_obj)
      green_obj = green(dos_obj)
      call energy_obj%e_mesh()
      call recursion_obj%zsqr()
      nen = energy_obj%channels_ldos + 10
      allocate (a_inf(18, 18, lattice_obj%nrec), b_inf(18, 18, lattice_obj%nrec), a_inf0(lattice_obj%nrec), b_inf0(lattice_obj%nrec))
      nw = 10*control_obj%lld
      call system_clock(tc0, trate)
      call recursion_obj%get_terminf(recursion_obj%a_b, recursion_obj%b2_b, atoms_per_process, control_obj%lld, 18, nw, a_inf, b_inf, a_inf0, b_inf0)
      call system_clock(tc1)
      call green_obj%block_green()
      call system_clock(tc2)
      write (*, '(a, f9.4, a, f9.4, a, i4, a)') ' dump_fixture: get_terminf ', real(tc1 - tc0)/real(trate), ' s, block_green (incl. its own get_terminf) ', &
         real(tc2 - tc1)/real(trate), ' s for ', lattice_obj%nrec, ' site(s)'

      sym_i = 0
      if (control_obj%sym_term) sym_i = 1
      write (u) int(z'47524e31'), nen, sym_i
      write (u) energy_obj%ene(1:nen)
      write (u) a_inf
      write (u) b_inf
      write (u) recursion_obj%b2_b
      write (u) green_obj%g0(:, :, 1:nen, 1:lattice_obj%nrec)
      ! ---- green%block_green_eta (green.f90:544-579): bgreen at single energy points with a complex energy increment eta
      !      (the Gauss-Legendre contour of the exchange code); NETA (point, eta) pairs, g_ef(18,18,site) each
      write (u) int(z'47524e33'), neta, 0
      allocate (g_ef(18, 18, atoms_per_process))
      do ieta = 1, neta
         fpt = 1 + ((nen - 1)*(ieta - 1))/(neta - 1)
         eta_c = cmplx(0.002_rp*mod(ieta, 3), 0.004_rp*ieta, rp)
         g_ef = (0.0_rp, 0.0_rp)
         call green_obj%block_green_eta(eta_c, fpt, g_ef)
         write (u) fpt, eta_c
         write (u) g_ef(:, :, 1:lattice_obj%nrec)
      end do
   case (1)
      write (u) recursion_obj%mu_n
      ! ---- the stage right after the Chebyshev recursion (self.f90:824): green%chebyshev_green (green.f90:1030-1108)
      dos_obj = dos(recursion_obj, energy_obj)
      green_obj = green(dos_obj)
      call energy_obj%e_mesh()
      call green_obj%chebyshev_green()
      nen = energy_obj%channels_ldos + 10
      write (u) int(z'47524e32'), nen, 0
      write (u) energy_obj%ene(1:nen)
      write (u) green_obj%g0(:, :, 1:nen, 1:lattice_obj%nrec)
   case (2)
      write (u) recursion_obj%a(:, :, :, 1)
      write (u) recursion_obj%b2(:, :, :, 1)
      ! ---- the stage right after the scalar recursion (self.f90:822-823, run_dos): green%sgreen (green.f90:628-705), which calls
      !      dos%density (density_of_states.f90:248-363: bpOPT band edges + one scalar continued fraction bprldos :370-404 per
      !      orbital and energy).  Appended so that the device version of that stage can be pinned: energies, the potential
      !      parameters density reads, its output for every (site, direction), and the g0 sgreen makes of it.
      dos_obj = dos(recursion_obj, energy_obj)
      green_obj = green(dos_obj)
      call energy_obj%e_mesh()
      nen = energy_obj%channels_ldos + 10
      write (u) int(z'44454e31'), nen, control_obj%nmdir
      write (u) energy_obj%ene(1:nen)
      do ia = 1, lattice_obj%nrec
         write (u) lattice_obj%symbolic_atoms(l2g_map(ia))%potential%dw_l(1:18)
         write (u) lattice_obj%symbolic_atoms(l2g_map(ia))%potential%cshi(1:18)
      end do
      write (u) recursion_obj%a(:, :, 1:lattice_obj%nrec, 1:control_obj%nmdir)
      write (u) recursion_obj%b2(:, :, 1:lattice_obj%nrec, 1:control_obj%nmdir)
      allocate (doso(18, nen))
      do sym_i = 1, control_obj%nmdir
         do ia = 1, lattice_obj%nrec
            doso = 0.0_rp
            call dos_obj%density(doso, ia, sym_i)
            write (u) doso
         end do
      end do
      call green_obj%sgreen()
      write (u) green_obj%g0(:, :, 1:nen, 1:lattice_obj%nrec)
   end select
   close (u)
   ! ---- local_axis = T (recursion.f90:1830-1832): the chain of site i ran on the blocks rotated into the spin frame of that
   !      site's moment (hamiltonian.f90:2442-2465); the arrays dumped above are in the LAST site's frame.  The global-frame
   !      blocks the rotation starts from, the moment directions and the rotation matrices ROTMAT(car2sph(mom)) (math.f90:2026)
   !      go to a side file.
   if (hamiltonian_obj%local_axis) call dump_local_axis()
   call get_environment_variable('RSREC_DUMP_HMAG', envbuf)
   if (len_trim(envbuf) > 0 .and. .not. hamiltonian_obj%local_axis) call dump_hmag()
   write (*, *) 'dump_fixture: wrote fixture.bin kk=', lattice_obj%kk, ' nmax=', lattice_obj%nmax, ' nrec=', lattice_obj%nrec, &
      ' lld=', control_obj%lld, ' nslots=', nslots, ' kind=', kind_rec, ' hoh=', hoh_i
contains
   ! ---- the INPUTS of build_bulkham / build_locham (hamiltonian.f90:1553-1667): the four 9x9 parts (Hx, Hy, Hz, H0) that chbar_nc
   !      leaves in hmag for every class atom, the type behind every neighbour slot, and obarm -- what a device-side assembly of
   !      ee / eeo / hall / hallo starts from (SURVEY 8 f2).  Their outputs are the blocks already in fixture.bin.
   subroutine dump_hmag()
      integer :: v, nt, ia_c, ino, nr, m, ja, nsl
      integer, allocatable :: ji(:)
      nsl = size(hamiltonian_obj%ee, 3)
      allocate (ji(nsl))
      open (newunit=v, file='hmag.bin', access='stream', form='unformatted', status='replace')
      write (v) int(z'484d4731'), lattice_obj%ntype, lattice_obj%nmax, nsl, hoh_i
      do nt = 1, lattice_obj%ntype
         ia_c = lattice_obj%atlist(nt)
         call one_class(ia_c, nt)
      end do
      if (control_obj%calctype == 'I') then
         do nt = 1, lattice_obj%nmax
            call one_class(nt, nt)
         end do
      end if
      write (v) hamiltonian_obj%obarm
      close (v)
   contains
      subroutine one_class(iatom, iclass)
         integer, intent(in) :: iatom, iclass
         ino = lattice_obj%num(iatom)
         nr = lattice_obj%nn(iatom, 1)
         call hamiltonian_obj%chbar_nc(iatom, nr, ino, iclass)
         ji = 0
         do m = 1, nr
            if (m == 1) then
               ji(m) = lattice_obj%iz(iatom)
            else
               ja = lattice_obj%nn(iatom, m)
               if (ja /= 0) ji(m) = lattice_obj%iz(ja)
            end if
         end do
         write (v) nr
         write (v) ji
         write (v) hamiltonian_obj%hmag(:, :, 1:nsl, 1:4)
      end subroutine one_class
   end subroutine dump_hmag

   subroutine dump_local_axis()
      use math_mod, only: car2sph, ROTMAT
      integer :: v, is
      real(rp) :: sv(3), mom(3)
      complex(rp) :: rmat(18, 18)
      open (newunit=v, file='local_axis.bin', access='stream', form='unformatted', status='replace')
      write (v) int(z'4c415831'), lattice_obj%nrec
      write (v) hamiltonian_obj%ee_glob
      if (hamiltonian_obj%hoh) then
         write (v) hamiltonian_obj%eeo_glob
         write (v) hamiltonian_obj%enim_glob
      end if
      if (lattice_obj%nmax > 0) then
         write (v) hamiltonian_obj%hall_glob
         if (hamiltonian_obj%hoh) write (v) hamiltonian_obj%hallo_glob
      end if
      do is = 1, lattice_obj%nrec
         mom = lattice_obj%symbolic_atoms(is)%potential%mom
         call car2sph(mom, sv)
         rmat = (0.0_rp, 0.0_rp)
         call ROTMAT(rmat, sv(1), sv(2), 0.0_rp)
         write (v) mom
         write (v) rmat
      end do
      close (v)
   end subroutine dump_local_axis
end program dump_fixture
