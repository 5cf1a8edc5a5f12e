!------------------------------------------------------------------------------
! recursion_gpu_mod -- MI355X drop-in for the recursion drivers of RS-LMTO-ASA.
!
! `type(recursion_gpu)` EXTENDS the reference's `type(recursion)` (source/recursion.f90:41-116) and overrides
! exactly the hot-path type-bound procedures
!     recur_b (:1807)   recur_b_ij (:1655)   chebyshev_recur (:3057)   chebyshev_recur_ij (:2376)   recur (:3485)   zsqr (:1980)
!     get_terminf (:2092)   compute_moments_stochastic (:979)   ham_[hoh_]vec_matmul (:913/:785)   velo_[hoh_]vec_matmul (:587/:656)
! with calls into librsrec (hand-written HIP kernels behind the C ABI of include/rsrec.h).  Everything else --
! data members a, b2, a_b, b2_b, mu_n (same shapes, same index order), bpopt / emami / get_cinf (the scalar-chain terminator of
! density_of_states.f90), setup_kubo_operators, chebyshev_orbital_mod, restore_to_default -- is inherited unchanged, and every consumer
! in the reference holds a `class(recursion), pointer` (self.f90:64, green.f90:45, density_of_states.f90:44,
! bands.f90:61), so the SCF loop dispatches to the GPU drivers without being edited.  The only edit a maintainer makes
! is the declaration/constructor in calculation.f90 (see INTEGRATION.md):
!     type(recursion_gpu), target :: recursion_obj ;  recursion_obj = recursion_gpu(hamiltonian_obj, energy_obj)
!
! Device context: ONE module-level handle per process, created lazily on the first driver call.  Nothing device-side
! lives in the type, so the reference's "construct by intrinsic assignment from a function result + final ::
! destructor" pattern (calculation.f90:599, recursion.f90:115,150) cannot leave a dangling handle.
! MPI rank r uses GPU mod(r, device_count); sites are split by the reference's own get_mpi_variables (mpi.f90:32).
! Errors: any non-zero status of the library becomes g_logger%fatal, the reference's error behaviour on this path
! (recursion.f90:1942 'Diagonalization error', :2595 'Chebyshev moments did not converge').
!------------------------------------------------------------------------------
module recursion_gpu_mod
   use, intrinsic :: iso_c_binding
   use mpi_mod
   use hamiltonian_mod
   use energy_mod
   use recursion_mod
   use precision_mod, only: rp
   use math_mod, only: one_over_sqrt_two, car2sph, ROTMAT
   use string_mod, only: int2str
   use logger_mod, only: g_logger
   use timer_mod, only: g_timer
   use rsrec_binding
   use rsrec_context_mod
   implicit none

   private

   type, public, extends(recursion) :: recursion_gpu
   contains
      procedure :: recur => gpu_recur
      procedure :: recur_b => gpu_recur_b
      procedure :: recur_b_ij => gpu_recur_b_ij
      procedure :: chebyshev_recur => gpu_chebyshev_recur
      procedure :: chebyshev_recur_ij => gpu_chebyshev_recur_ij
      procedure :: zsqr => gpu_zsqr
      procedure :: get_terminf => gpu_get_terminf
      procedure :: compute_moments_stochastic => gpu_compute_moments_stochastic
      procedure :: chebyshev_orbital_mod => gpu_chebyshev_orbital_mod
      procedure :: ham_vec_matmul => gpu_ham_vec_matmul
      procedure :: ham_hoh_vec_matmul => gpu_ham_hoh_vec_matmul
      procedure :: velo_vec_matmul => gpu_velo_vec_matmul
      procedure :: velo_hoh_vec_matmul => gpu_velo_hoh_vec_matmul
   end type recursion_gpu

   interface recursion_gpu
      procedure :: gpu_constructor
   end interface recursion_gpu

   public :: rsrec_gpu_shutdown, rsrec_gpu_context, rsrec_gpu_block_resident

   !> (the per-process device context g_handle lives in rsrec_context_mod; re-exported above for the hosts that used it from here)
   !> number of sites whose block coefficients the last driver call left on the device (0: none -- another driver ran since, or the
   !> coefficients there are not the ones in a_b / b2_b, as after a local-axis run): the input of the device LDOS stage
   integer, save :: g_block_resident = 0

contains

   !> Same construction as recursion.f90:132-143 (pointers + restore_to_default); no device state is created here.
   function gpu_constructor(hamiltonian_obj, energy_obj) result(obj)
      type(recursion_gpu) :: obj
      class(hamiltonian), target, intent(in) :: hamiltonian_obj     ! (class: a type(hamiltonian_gpu) is accepted as well)
      type(energy), target, intent(in) :: energy_obj

      obj%hamiltonian => hamiltonian_obj
      obj%lattice => hamiltonian_obj%charge%lattice
      obj%en => energy_obj
      obj%control => hamiltonian_obj%charge%lattice%control
      call obj%restore_to_default()
   end function gpu_constructor

   !> Sites of this rank whose a_b / b2_b (as recur_b produced them) are also resident on the device; 0 if they are not.
   function rsrec_gpu_block_resident() result(n)
      integer :: n
      n = g_block_resident
   end function rsrec_gpu_block_resident

   subroutine check(rc, where)
      integer(c_int), intent(in) :: rc
      character(len=*), intent(in) :: where
      if (rc /= 0) call g_logger%fatal(where//': '//rsrec_error_string(g_handle), __FILE__, __LINE__)
   end subroutine check

   !> Create the context on first use and (re)send the tables the recursion reads.
   !> The caller rebuilds the Hamiltonian before every recur* call (self.f90:777-797), so the blocks are uploaded
   !> on every call; the lattice tables are constant during a run but small (kk*(nnmax+1) int32).
   subroutine sync_device(this, upload_lattice)
      class(recursion_gpu), intent(inout), target :: this
      logical, intent(in) :: upload_lattice
      integer(c_int) :: rc, hoh_i
      type(c_ptr) :: p_hall, p_hallo, ctx

      ctx = rsrec_gpu_context()                                  ! creates g_handle on first use
      if (upload_lattice) then
         rc = rsrec_set_lattice(g_handle, int(this%lattice%kk, c_int), int(size(this%lattice%nn, 2), c_int), &
                                c_loc(this%lattice%nn), c_loc(this%lattice%iz), int(this%lattice%nmax, c_int), &
                                int(this%lattice%ntype, c_int))
         call check(rc, 'rsrec_set_lattice')
         ! locality hint only (processing order of the atoms); the arithmetic never reads the positions
         if (allocated(this%lattice%cr)) then
            if (size(this%lattice%cr, 1) == 3 .and. size(this%lattice%cr, 2) >= this%lattice%kk) then
               rc = rsrec_set_positions(g_handle, c_loc(this%lattice%cr))
               call check(rc, 'rsrec_set_positions')
            end if
         end if
      end if
      hoh_i = 0
      if (this%hamiltonian%hoh) hoh_i = 1
      p_hall = c_null_ptr
      p_hallo = c_null_ptr
      if (this%lattice%nmax > 0) then
         p_hall = c_loc(this%hamiltonian%hall)
         p_hallo = c_loc(this%hamiltonian%hallo)
      end if
      rc = rsrec_set_hamiltonian(g_handle, int(size(this%hamiltonian%ee, 3), c_int), hoh_i, int(this%control%nsp, c_int), &
                                 c_loc(this%hamiltonian%ee), c_loc(this%hamiltonian%lsham), c_loc(this%hamiltonian%eeo), &
                                 c_loc(this%hamiltonian%enim), p_hall, p_hallo)
      call check(rc, 'rsrec_set_hamiltonian')
   end subroutine sync_device

   !> local_axis runs: the library takes the GLOBAL-frame blocks (hamiltonian%*_glob) once; lsham is frame-independent input.
   subroutine sync_device_global_frame(this)
      class(recursion_gpu), intent(inout), target :: this
      integer(c_int) :: rc, hoh_i
      type(c_ptr) :: p_hall, p_hallo, p_eeo, p_enim
      call sync_device(this, .true.)                             ! context + lattice tables (the blocks sent here are replaced below)
      hoh_i = 0
      p_eeo = c_null_ptr; p_enim = c_null_ptr; p_hall = c_null_ptr; p_hallo = c_null_ptr
      if (this%hamiltonian%hoh) then
         hoh_i = 1
         p_eeo = c_loc(this%hamiltonian%eeo_glob)
         p_enim = c_loc(this%hamiltonian%enim_glob)
      end if
      if (this%lattice%nmax > 0) then
         p_hall = c_loc(this%hamiltonian%hall_glob)
         if (this%hamiltonian%hoh) p_hallo = c_loc(this%hamiltonian%hallo_glob)
      end if
      rc = rsrec_set_hamiltonian(g_handle, int(size(this%hamiltonian%ee_glob, 3), c_int), hoh_i, int(this%control%nsp, c_int), &
                                 c_loc(this%hamiltonian%ee_glob), c_loc(this%hamiltonian%lsham), p_eeo, p_enim, p_hall, p_hallo)
      call check(rc, 'rsrec_set_hamiltonian')
   end subroutine sync_device_global_frame

   !---------------------------------------------------------------------------
   !> Block recursion for the sites of this rank (replaces recursion.f90:1807-1866)
   !---------------------------------------------------------------------------
   subroutine gpu_recur_b(this)
      class(recursion_gpu), intent(inout) :: this
      integer :: i, j, l, ll, llmax, nloc, i_loc
      integer(c_int) :: rc
      integer(c_int), allocatable, target :: seeds(:)
      complex(rp), allocatable, target :: ab(:, :, :, :), bb(:, :, :, :), rot(:, :, :)
      real(rp) :: mom(3), sv(3)

      call get_mpi_variables(rank, this%lattice%nrec)            ! recursion.f90:1816
      llmax = this%lattice%control%lld
      nloc = end_atom - start_atom + 1
      g_block_resident = 0
      if (nloc <= 0) return
      allocate (seeds(nloc), ab(18, 18, llmax, nloc), bb(18, 18, llmax, nloc))

      if (this%hamiltonian%local_axis) then
         ! The reference re-rotates every block into the spin frame of each site before its chain (recursion.f90:1830-1832), i.e.
         ! H is per site.  All sites still go in ONE call: the library gets the GLOBAL-frame blocks (the *_glob arrays
         ! rotate_to_local_axis starts from, hamiltonian.f90:2451-2461) plus one rotation matrix per site, runs every chain on the
         ! global blocks with that site's on-site term R l.s R^H and conjugates the 18x18 outputs (see include/rsrec.h).
         allocate (rot(18, 18, nloc))
         do i = start_atom, end_atom
            i_loc = i - start_atom + 1
            j = this%lattice%irec(i)
            call g_logger%info('Block recursion on progress for atom '//int2str(j), __FILE__, __LINE__)
            seeds(i_loc) = int(j, c_int)
            mom = this%lattice%symbolic_atoms(i)%potential%mom
            call car2sph(mom, sv)                                 ! rotmag_loc, math.f90:2021-2026
            rot(:, :, i_loc) = (0.0_rp, 0.0_rp)
            call ROTMAT(rot(:, :, i_loc), sv(1), sv(2), 0.0_rp)
         end do
         call sync_device_global_frame(this)
         call g_timer%start('H|PSI_n>')
         rc = rsrec_block_lanczos_local_axis(g_handle, int(nloc, c_int), c_loc(seeds), c_loc(rot), int(llmax, c_int), c_loc(ab), c_loc(bb))
         call g_timer%stop('H|PSI_n>')
         call check(rc, 'rsrec_block_lanczos_local_axis')
         ! leave the Hamiltonian object as the reference's loop does: rotated into the frame of the last site
         call this%hamiltonian%rotate_to_local_axis(this%lattice%symbolic_atoms(end_atom)%potential%mom)
      else
         call sync_device(this, .true.)
         do i = start_atom, end_atom
            j = this%lattice%irec(i)
            call g_logger%info('Block recursion on progress for atom '//int2str(j), __FILE__, __LINE__)
            seeds(i - start_atom + 1) = int(j, c_int)
         end do
         call g_timer%start('H|PSI_n>')       ! the reference's label for the hot loop (recursion.f90:1902); here: all levels of all sites
         rc = rsrec_block_lanczos(g_handle, int(nloc, c_int), c_loc(seeds), int(llmax, c_int), c_loc(ab), c_loc(bb))
         call g_timer%stop('H|PSI_n>')
         call check(rc, 'rsrec_block_lanczos')
         g_block_resident = nloc
      end if

      do i_loc = 1, nloc                                          ! recursion.f90:1844-1853
         this%a_b(:, :, 1:llmax, i_loc) = ab(:, :, :, i_loc)
         this%b2_b(:, :, 1:llmax, i_loc) = bb(:, :, :, i_loc)
         do ll = 1, llmax
            do l = 1, 18
               this%a(ll, l, i_loc, 1) = real(ab(l, l, ll, i_loc))
               this%b2(ll, l, i_loc, 1) = real(bb(l, l, ll, i_loc))
            end do
         end do
      end do
      ! debug dump kept for compatibility with existing tooling (recursion.f90:1856-1865)
      do i = start_atom, end_atom
         do l = 1, 18
            write (1000*(1 + rank) + 122, *) 'orbital', l, 'atom', i
            do ll = 1, llmax
               write (1000*(1 + rank) + 122, '(2f12.8,4x,2f12.8)') this%a(ll, l, i - start_atom + 1, 1), this%b2(ll, l, i - start_atom + 1, 1)
            end do
         end do
      end do
   end subroutine gpu_recur_b

   !---------------------------------------------------------------------------
   !> Four chains per atom pair (replaces recursion.f90:1655-1737)
   !---------------------------------------------------------------------------
   subroutine gpu_recur_b_ij(this)
      class(recursion_gpu), intent(inout) :: this
      integer :: i, j, ij, ij_loc, reci, llmax, nch, c
      integer(c_int) :: rc
      integer(c_int), allocatable, target :: seeds(:, :)
      integer, allocatable :: slot(:)
      complex(rp), allocatable, target :: coef(:, :), ab(:, :, :, :), bb(:, :, :, :)

      llmax = this%lattice%control%lld
      allocate (seeds(2, 4*max(end_atom - start_atom + 1, 1)), coef(2, 4*max(end_atom - start_atom + 1, 1)), slot(4*max(end_atom - start_atom + 1, 1)))
      nch = 0
      do ij = start_atom, end_atom
         ij_loc = g2l_map(ij)
         i = this%lattice%ijpair(ij, 1)
         j = this%lattice%ijpair(ij, 2)
         call g_logger%info('Block recursion on progress between atoms '//int2str(i)//' and '//int2str(j), __FILE__, __LINE__)
         do reci = 1, 4
            if (i == j .and. reci > 1) cycle                       ! :1705-1706
            nch = nch + 1
            seeds(:, nch) = [int(i, c_int), int(j, c_int)]
            slot(nch) = ij_loc*4 - 4 + reci                        ! :1721
            if (i == j) then
               coef(:, nch) = [(1.0_rp, 0.0_rp), (1.0_rp, 0.0_rp)] ! asign = bsign = 1 (:1702-1704); seeds are assigned in order
            else
               coef(1, nch) = (1.0_rp, 0.0_rp)*one_over_sqrt_two
               select case (reci)                                  ! :1679-1700
               case (1); coef(2, nch) = (1.0_rp, 0.0_rp)*one_over_sqrt_two
               case (2); coef(2, nch) = (-1.0_rp, 0.0_rp)*one_over_sqrt_two
               case (3); coef(2, nch) = (0.0_rp, 1.0_rp)*one_over_sqrt_two
               case (4); coef(2, nch) = (0.0_rp, -1.0_rp)*one_over_sqrt_two
               end select
            end if
         end do
      end do
      if (nch == 0) return
      allocate (ab(18, 18, llmax, nch), bb(18, 18, llmax, nch))
      call sync_device(this, .true.)
      g_block_resident = 0
      rc = rsrec_block_lanczos_seeded(g_handle, int(nch, c_int), 2_c_int, c_loc(seeds), c_loc(coef), int(llmax, c_int), c_loc(ab), c_loc(bb))
      call check(rc, 'rsrec_block_lanczos_seeded')
      do c = 1, nch
         this%a_b(:, :, 1:llmax, slot(c)) = ab(:, :, :, c)
         this%b2_b(:, :, 1:llmax, slot(c)) = bb(:, :, :, c)
      end do
   end subroutine gpu_recur_b_ij

   !---------------------------------------------------------------------------
   !> b2_b <- sqrt(b2_b) for every level of every local chain (replaces recursion.f90:1980-2023)
   !---------------------------------------------------------------------------
   subroutine gpu_zsqr(this)
      class(recursion_gpu), intent(inout) :: this
      integer :: na
      integer(c_int) :: rc
      type(c_ptr) :: ctx
      complex(rp), allocatable, target :: buf(:, :, :, :)

      if (this%lattice%njij == 0) then
         na = atoms_per_process
      else
         na = atoms_per_process*4
      end if
      if (na <= 0) return
      ctx = rsrec_gpu_context()                                  ! (the square roots read no lattice or operator table: nothing to send)
      allocate (buf(18, 18, this%lattice%control%lld, na))
      buf = this%b2_b(:, :, 1:this%lattice%control%lld, 1:na)
      rc = rsrec_zsqr(g_handle, int(this%lattice%control%lld*na, c_int), c_loc(buf))
      call check(rc, 'rsrec_zsqr')
      this%b2_b(:, :, 1:this%lattice%control%lld, 1:na) = buf
   end subroutine gpu_zsqr

   !---------------------------------------------------------------------------
   !> Chebyshev moments for the sites of this rank (replaces recursion.f90:3057-3130)
   !---------------------------------------------------------------------------
   subroutine gpu_chebyshev_recur(this)
      class(recursion_gpu), intent(inout) :: this
      integer :: i, j, nloc, nmom
      integer(c_int) :: rc
      real(rp) :: a, b
      integer(c_int), allocatable, target :: seeds(:)
      complex(rp), allocatable, target :: mu(:, :, :, :)

      nloc = end_atom - start_atom + 1
      if (nloc <= 0) return
      nmom = 2*this%control%lld + 2
      a = (this%en%energy_max - this%en%energy_min)/(2 - 0.3)      ! recursion.f90:3078 (0.3 is a default-real literal there too)
      b = (this%en%energy_max + this%en%energy_min)/2
      allocate (seeds(nloc), mu(18, 18, nmom, nloc))
      do i = start_atom, end_atom
         j = this%lattice%irec(i)
         call g_logger%info('Chebyshev recursion on progress for atom '//int2str(j), __FILE__, __LINE__)
         seeds(g2l_map(i)) = int(j, c_int)
      end do
      call sync_device(this, .true.)
      call g_timer%start('<PSI_0|PSI_n>')
      g_block_resident = 0
      rc = rsrec_chebyshev(g_handle, int(nloc, c_int), c_loc(seeds), int(this%control%lld, c_int), real(a, c_double), real(b, c_double), c_loc(mu))
      call g_timer%stop('<PSI_0|PSI_n>')
      call check(rc, 'rsrec_chebyshev')
      this%mu_n(:, :, 1:nmom, 1:nloc) = mu
   end subroutine gpu_chebyshev_recur

   !---------------------------------------------------------------------------
   !> Chebyshev moments of the four chains per atom pair (replaces recursion.f90:2376-2487)
   !---------------------------------------------------------------------------
   subroutine gpu_chebyshev_recur_ij(this)
      class(recursion_gpu), intent(inout) :: this
      integer :: i, j, ij, ij_loc, reci, nch, c, nmom
      integer(c_int) :: rc
      real(rp) :: a, b
      integer(c_int), allocatable, target :: seeds(:, :)
      integer, allocatable :: slot(:)
      complex(rp), allocatable, target :: coef(:, :), mu(:, :, :, :)

      nch = 4*max(end_atom - start_atom + 1, 0)
      if (nch <= 0) return
      nmom = 2*this%control%lld + 2
      a = (this%en%energy_max - this%en%energy_min)/(2 - 0.3)
      b = (this%en%energy_max + this%en%energy_min)/2
      allocate (seeds(2, nch), coef(2, nch), slot(nch), mu(18, 18, nmom, nch))
      c = 0
      do ij = start_atom, end_atom
         ij_loc = g2l_map(ij)
         i = this%lattice%ijpair(ij, 1)
         j = this%lattice%ijpair(ij, 2)
         call g_logger%info(int2str(rank)//': Chebyshev recursion on progress between atoms '//int2str(i)//' and '//int2str(j), __FILE__, __LINE__)
         do reci = 1, 4                                            ! no i == j special case in the reference (:2403-2448)
            c = c + 1
            seeds(:, c) = [int(i, c_int), int(j, c_int)]
            slot(c) = ij_loc*4 - 4 + reci
            coef(1, c) = (1.0_rp, 0.0_rp)*one_over_sqrt_two
            select case (reci)
            case (1); coef(2, c) = (1.0_rp, 0.0_rp)*one_over_sqrt_two
            case (2); coef(2, c) = (-1.0_rp, 0.0_rp)*one_over_sqrt_two
            case (3); coef(2, c) = (0.0_rp, 1.0_rp)*one_over_sqrt_two
            case (4); coef(2, c) = (0.0_rp, -1.0_rp)*one_over_sqrt_two
            end select
         end do
      end do
      call sync_device(this, .true.)
      call g_timer%start('<PSI_0|PSI_n>')
      g_block_resident = 0
      rc = rsrec_chebyshev_seeded(g_handle, int(nch, c_int), 2_c_int, c_loc(seeds), c_loc(coef), int(this%control%lld, c_int), &
                                  real(a, c_double), real(b, c_double), c_loc(mu))
      call g_timer%stop('<PSI_0|PSI_n>')
      call check(rc, 'rsrec_chebyshev_seeded')
      do c = 1, nch
         this%mu_n(:, :, 1:nmom, slot(c)) = mu(:, :, :, c)
      end do
   end subroutine gpu_chebyshev_recur_ij

   !---------------------------------------------------------------------------
   !> Scalar Haydock recursion (replaces recursion.f90:3485-3532)
   !---------------------------------------------------------------------------
   subroutine gpu_recur(this)
      class(recursion_gpu), intent(inout) :: this
      integer :: i, nloc, llmax_a
      integer(c_int) :: rc
      integer(c_int), allocatable, target :: seeds(:)
      real(rp), allocatable, target :: a(:, :, :), b2(:, :, :)

      nloc = end_atom - start_atom + 1
      if (nloc <= 0) return
      llmax_a = size(this%a, 1)
      allocate (seeds(nloc), a(llmax_a, 18, nloc), b2(llmax_a, 18, nloc))
      do i = start_atom, end_atom
         seeds(g2l_map(i)) = int(this%lattice%irec(i), c_int)
      end do
      call sync_device(this, .true.)
      g_block_resident = 0
      rc = rsrec_scalar_lanczos(g_handle, int(nloc, c_int), c_loc(seeds), int(this%lattice%control%lld, c_int), int(llmax_a, c_int), c_loc(a), c_loc(b2))
      call check(rc, 'rsrec_scalar_lanczos')
      ! the reference fills rows 1..lld of a(:,:,i_loc,1) / b2 and leaves the rest untouched (:3516-3519)
      this%a(1:this%lattice%control%lld, :, 1:nloc, 1) = a(1:this%lattice%control%lld, :, :)
      this%b2(1:this%lattice%control%lld, :, 1:nloc, 1) = b2(1:this%lattice%control%lld, :, :)
   end subroutine gpu_recur

   !---------------------------------------------------------------------------
   !> Band-dependent terminator (replaces recursion.f90:2092-2135 with get_cinf :2030, bpopt :3540, emami :3589): all `na` sites and
   !> all 324 matrix elements in one kernel launch
   !---------------------------------------------------------------------------
   subroutine gpu_get_terminf(this, Acoef_b, B2coef_b, na, ll, ldim, nw, a_inf, b_inf, a_inf0, b_inf0)
      class(recursion_gpu), intent(inout) :: this
      integer, intent(in) :: na
      integer, intent(in) :: ll
      integer, intent(inout) :: nw
      integer, intent(in) :: ldim
      complex(rp), dimension(ldim, ldim, ll, na), intent(in) :: Acoef_b, B2coef_b
      real(rp), dimension(ldim, ldim, na), intent(out) :: a_inf, b_inf
      real(rp), dimension(na), intent(out) ::  a_inf0, b_inf0
      integer(c_int) :: rc
      complex(rp), allocatable, target :: ac(:, :, :, :), bc(:, :, :, :)
      real(rp), allocatable, target :: ai(:, :, :), bi(:, :, :), a0(:), b0(:)
      type(c_ptr) :: h

      if (ldim /= 18) call g_logger%fatal('recursion_gpu%get_terminf: ldim must be 18', __FILE__, __LINE__)
      if (na <= 0) return
      h = rsrec_gpu_context()
      allocate (ac(18, 18, ll, na), bc(18, 18, ll, na), ai(18, 18, na), bi(18, 18, na), a0(na), b0(na))
      ac = Acoef_b; bc = B2coef_b
      rc = rsrec_terminator(h, int(na, c_int), int(ll, c_int), c_loc(ac), c_loc(bc), c_loc(ai), c_loc(bi), c_loc(a0), c_loc(b0))
      call check(rc, 'rsrec_terminator')
      a_inf = ai; b_inf = bi; a_inf0 = a0; b_inf0 = b0
   end subroutine gpu_get_terminf

   !---------------------------------------------------------------------------
   !> Stochastic Kubo-Bastin double moments (replaces recursion.f90:979-1234).  Operator set-up (setup_kubo_operators :242) and the
   !> seeds stay the reference's: per_type -> identity on lattice%atlist(i); random_vec -> the reference's random_seed / random_number
   !> sequence, one phase per atom (:1103-1114).  Everything after that -- 3 cond_ll block SpMMs and the cond_ll^2 moment contraction
   !> per vector -- runs in one library call.
   !---------------------------------------------------------------------------
   subroutine gpu_compute_moments_stochastic(this)
      use math_mod, only: pi, i_unit
      class(recursion_gpu), intent(inout) :: this
      integer :: i, k, loop_over, nseed, cll
      integer(c_int) :: rc
      real(rp) :: a, b, rng
      integer(c_int), allocatable, target :: seeds(:, :)
      complex(rp), allocatable, target :: coef(:, :), mu(:, :, :, :, :), va(:, :, :, :), vb(:, :, :, :), voa(:, :, :, :), vob(:, :, :, :)
      type(c_ptr) :: p_voa, p_vob

      cll = this%control%cond_ll
      select case (this%control%cond_calctype)
      case ('per_type')
         loop_over = this%lattice%ntype
         nseed = 1
      case ('random_vec')
         loop_over = this%control%random_vec_num
         nseed = this%lattice%kk
      case default
         call g_logger%fatal('recursion_gpu%compute_moments_stochastic: unknown cond_calctype', __FILE__, __LINE__)
         return
      end select
      if (allocated(this%mu_nm_stochastic)) deallocate (this%mu_nm_stochastic)
      allocate (this%mu_nm_stochastic(18, 18, cll, cll, loop_over))
      a = (this%en%energy_max - this%en%energy_min)/(2 - 0.3)      ! :1023-1024
      b = (this%en%energy_max + this%en%energy_min)/2
      call this%setup_kubo_operators(this%control%linear_out, this%control%linear_in)
      allocate (seeds(nseed, loop_over), coef(nseed, loop_over))
      do i = 1, loop_over
         call random_seed()                                       ! :1076
         if (nseed == 1) then
            seeds(1, i) = int(this%lattice%atlist(i), c_int)
            coef(1, i) = (1.0_rp, 0.0_rp)
         else
            do k = 1, this%lattice%kk
               call random_number(rng)
               seeds(k, i) = int(k, c_int)
               coef(k, i) = exp(2.0_rp*pi*i_unit*rng)/sqrt(real(this%lattice%kk))
            end do
         end if
      end do
      call sync_device(this, .true.)
      p_voa = c_null_ptr; p_vob = c_null_ptr
      allocate (va, source=this%hamiltonian%v_a)
      allocate (vb, source=this%hamiltonian%v_b)
      if (this%hamiltonian%hoh) then
         allocate (voa, source=this%hamiltonian%vo_a)
         allocate (vob, source=this%hamiltonian%vo_b)
         p_voa = c_loc(voa)
         p_vob = c_loc(vob)
      end if
      allocate (mu(18, 18, cll, cll, loop_over))
      rc = rsrec_kubo_moments(g_handle, int(loop_over, c_int), int(nseed, c_int), c_loc(seeds), c_loc(coef), int(cll, c_int), &
                              real(a, c_double), real(b, c_double), c_loc(va), p_voa, c_loc(vb), p_vob, c_loc(mu))
      call check(rc, 'rsrec_kubo_moments')
      this%mu_nm_stochastic = mu
   end subroutine gpu_compute_moments_stochastic

   !---------------------------------------------------------------------------
   !> Orbital moment from position-operator Chebyshev moments (replaces recursion.f90:2834-3049).  The reference loops over all kk
   !> atoms as seeds and sends every whole-lattice product through `ham_vec_matmul` on host arrays; here the seeds are chains of ONE
   !> library call (rsrec_orbital_moments: all vectors resident on the device, the moment sums reduced there).  What follows the
   !> moments -- the 1/kk average, the Jackson kernel, the energy sum, trace and integral written to unit 50 (:3006-3047) -- is the
   !> reference's, restated.  Not reproduced: the per-seed diagnostic prints (:2909, :2972), and the reference's accumulation onto a
   !> never-zeroed mu_n_orb (:2907 is commented out) -- the sum starts from zero here.
   !---------------------------------------------------------------------------
   subroutine gpu_chebyshev_orbital_mod(this)
      use math_mod, only: jackson_kernel, rtrace, simpson_f, i_unit, pi
      class(recursion_gpu), intent(inout) :: this
      integer :: i, l, m, ie, nv, ll, k
      integer(c_int) :: rc
      integer(c_int), allocatable, target :: seeds(:)
      real(rp), allocatable, target :: cr(:, :)
      complex(rp), allocatable, target :: mu_n_orb(:, :, :)
      complex(rp), dimension(18, 18, this%en%channels_ldos + 10) :: g0
      real(rp), dimension(this%control%lld) :: kernel
      real(rp), dimension(this%en%channels_ldos + 10) :: wscale, lzi
      complex(rp) :: exp_factor
      real(rp) :: a, b, lz

      ll = this%control%lld
      nv = this%en%channels_ldos + 10
      a = (this%en%energy_max - this%en%energy_min)/(2 - 0.3)      ! :2869-2870 (default-REAL literals, as there)
      b = (this%en%energy_max + this%en%energy_min)/2
      wscale(:) = (this%en%ene(:) - b)/a
      call jackson_kernel(ll, kernel)
      call sync_device(this, .true.)
      g_block_resident = 0
      allocate (seeds(this%lattice%kk), cr(3, this%lattice%kk), mu_n_orb(18, 18, ll))
      do k = 1, this%lattice%kk
         seeds(k) = int(k, c_int)
      end do
      cr = this%lattice%cr(1:3, 1:this%lattice%kk)
      call g_timer%start('chebyshev-orbital-gpu')
      rc = rsrec_orbital_moments(g_handle, int(this%lattice%kk, c_int), c_loc(seeds), int(ll, c_int), real(a, c_double), real(b, c_double), &
                                 c_loc(cr), real(this%lattice%alat, c_double), c_loc(mu_n_orb), c_null_ptr)
      call g_timer%stop('chebyshev-orbital-gpu')
      call check(rc, 'rsrec_orbital_moments')
      this%izero(:) = 1

      mu_n_orb(:, :, :) = mu_n_orb(:, :, :)/real(this%lattice%kk)   ! :3006
      do l = 1, 18
         do m = 1, 18
            mu_n_orb(l, m, :) = mu_n_orb(l, m, :)*kernel(:)
         end do
      end do
      mu_n_orb(:, :, 2:size(kernel)) = mu_n_orb(:, :, 2:size(kernel))*2.0_rp
      g0(:, :, :) = (0.0d0, 0.0d0)
      do ie = 1, nv                                                 ! :3019-3035
         do i = 1, size(kernel)
            exp_factor = -i_unit*exp(-i_unit*(i - 1)*acos(wscale(ie)))
            g0(:, :, ie) = g0(:, :, ie) + mu_n_orb(:, :, i)*aimag(exp_factor)
         end do
         g0(:, :, ie) = g0(:, :, ie)/((sqrt((a**2) - ((this%en%ene(ie) - b)**2))))
      end do
      do ie = 1, nv
         lzi(ie) = rtrace(g0(:, :, ie))
      end do
      do ie = 1, nv                                                 ! :3043-3046
         call simpson_f(lz, this%en%ene, this%en%ene(ie), this%en%nv1, lzi, .true., .false., 0.0d0)
         write (50, '(3es16.6)') this%en%ene(ie) - this%en%fermi, -(lz/pi), -(1/pi)*lzi(ie)
      end do
   end subroutine gpu_chebyshev_orbital_mod

   !---------------------------------------------------------------------------
   !> Whole-vector products on caller arrays (replace recursion.f90:913, :785, :587, :656).  With these overridden, the inherited
   !> chebyshev_orbital_mod (:2834) runs its H|psi> on the GPU.  The region flags are left "all active" for the caller's
   !> `izero = idum` bookkeeping (blocks outside the reference's region are exact zeros).
   !---------------------------------------------------------------------------
   subroutine gpu_apply(this, vel, v_op, vo_op, psi_in, psi_out, a, b)
      class(recursion_gpu), intent(inout), target :: this
      integer, intent(in) :: vel
      complex(rp), dimension(:, :, :, :), intent(in), target, optional :: v_op, vo_op
      complex(rp), dimension(:, :, :), intent(in) :: psi_in
      complex(rp), dimension(:, :, :), intent(out) :: psi_out
      real(rp), intent(in) :: a, b
      integer(c_int) :: rc
      complex(rp), allocatable, target :: xin(:, :, :), xout(:, :, :), v(:, :, :, :), vo(:, :, :, :)
      type(c_ptr) :: pv, pvo
      call sync_device(this, .true.)
      allocate (xin(18, 18, this%lattice%kk), xout(18, 18, this%lattice%kk))
      xin = psi_in(:, :, 1:this%lattice%kk)
      pv = c_null_ptr; pvo = c_null_ptr
      if (present(v_op)) then
         allocate (v, source=v_op)
         pv = c_loc(v)
      end if
      if (present(vo_op)) then
         allocate (vo, source=vo_op)
         pvo = c_loc(vo)
      end if
      rc = rsrec_apply_operator(g_handle, int(vel, c_int), pv, pvo, c_loc(xin), c_loc(xout), real(a, c_double), real(b, c_double))
      call check(rc, 'rsrec_apply_operator')
      psi_out(:, :, 1:this%lattice%kk) = xout
      this%idum(:) = 1
   end subroutine gpu_apply

   subroutine gpu_ham_vec_matmul(this, psi_in, psi_out, a, b)
      class(recursion_gpu), intent(inout) :: this
      complex(rp), dimension(:, :, :), intent(in) :: psi_in
      complex(rp), dimension(:, :, :), intent(out) :: psi_out
      real(rp), intent(in) :: a, b
      call gpu_apply(this, 2, psi_in=psi_in, psi_out=psi_out, a=a, b=b)     ! 2: the plain operator ee + l.s also when hoh is set (:913-977)
   end subroutine gpu_ham_vec_matmul

   subroutine gpu_ham_hoh_vec_matmul(this, psi_in, psi_out, a, b)
      class(recursion_gpu), intent(inout) :: this
      complex(rp), dimension(:, :, :), intent(in) :: psi_in
      complex(rp), dimension(:, :, :), intent(out) :: psi_out
      real(rp), intent(in) :: a, b
      call gpu_apply(this, 0, psi_in=psi_in, psi_out=psi_out, a=a, b=b)     ! the library applies the operator that was set (hoh or not)
   end subroutine gpu_ham_hoh_vec_matmul

   subroutine gpu_velo_vec_matmul(this, c_or_n, v_op, psi_in, psi_out)
      class(recursion_gpu), intent(inout) :: this
      complex(rp), dimension(:, :, :, :), intent(in) :: v_op
      complex(rp), dimension(:, :, :), intent(in) :: psi_in
      character :: c_or_n
      complex(rp), dimension(:, :, :), intent(out) :: psi_out
      if (c_or_n /= 'n' .and. c_or_n /= 'N') call g_logger%fatal("recursion_gpu%velo_vec_matmul: only 'n' (every call site of the reference)", __FILE__, __LINE__)
      call gpu_apply(this, 1, v_op=v_op, psi_in=psi_in, psi_out=psi_out, a=1.0_rp, b=0.0_rp)
   end subroutine gpu_velo_vec_matmul

   subroutine gpu_velo_hoh_vec_matmul(this, v_op, vo_op, psi_in, psi_out)
      class(recursion_gpu), intent(inout) :: this
      complex(rp), dimension(:, :, :), intent(in) :: psi_in
      complex(rp), dimension(:, :, :), intent(out) :: psi_out
      complex(rp), dimension(:, :, :, :), intent(in) :: v_op
      complex(rp), dimension(:, :, :, :), intent(in) :: vo_op
      call gpu_apply(this, 1, v_op=v_op, vo_op=vo_op, psi_in=psi_in, psi_out=psi_out, a=1.0_rp, b=0.0_rp)
   end subroutine gpu_velo_hoh_vec_matmul

end module recursion_gpu_mod
