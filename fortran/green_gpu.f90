!------------------------------------------------------------------------------
! RS-LMTO-ASA drop-in, second stage: the Green function of the block recursion on the GPU.
!------------------------------------------------------------------------------
!
! MODULE: green_gpu_mod
!
! DESCRIPTION:
!> `type, extends(green) :: green_gpu` overrides the continued fraction of the reference's `green` type, `bgreen` (and its all-sites driver `block_green`, to batch the sites)
!> (green.f90:1191-1339), the continued fraction  coefficients -> g(E)  that `block_green` (:588), `block_green_eta`
!> (:541) and the inter-site variants call per site.  Everything else -- the terminator (`recursion%get_terminf`),
!> the result arrays `g0, gij, ...`, `sgreen`, `chebyshev_green` -- is inherited.  The GPU side is `rsrec_block_green`
!> (include/rsrec.h): one wave per energy point, register-resident 18x18 complex inverse with LAPACK's pivot rule (LDS carries only pivot rows / columns).
!>
!> `bands`, `self` and `exchange` hold `class(green), pointer` (bands.f90:49, self.f90:76, exchange.f90:48); the only
!> non-polymorphic spot is the dummy of the `bands` constructor (bands.f90:121, `type(green), target`), which a maintainer
!> changes to `class(green), target` (INTEGRATION.md).
!------------------------------------------------------------------------------
module green_gpu_mod
   use, intrinsic :: iso_c_binding
   use green_mod
   use density_of_states_mod, only: dos
   use precision_mod, only: rp
   use logger_mod, only: g_logger
   use timer_mod, only: g_timer
   use rsrec_binding
   use rsrec_context_mod, only: rsrec_gpu_context, rsrec_env_flag
   implicit none

   private

   type, public, extends(green) :: green_gpu
      !> `g0` on demand: with defer_g0 = .true. `block_green` only notes that `g0` is out of date; the continued fraction and the
      !> 13 MB per site it brings over PCIe happen in `fetch_g0`, which the consumers of `g0` call (bands_gpu does before every
      !> inherited routine that reads it).  A flow that only needs densities of states (`bands_gpu%calculate_fermi`, served by the
      !> device LDOS stage from the coefficients the recursion left on the GPU) then never produces `g0` at all.
      logical :: defer_g0 = .false.
      logical :: g0_stale = .false.
      logical :: fetching = .false.   ! set by fetch_g0 around its own block_green call: the one caller that makes a deferred g0
   contains
      procedure :: bgreen => gpu_bgreen
      procedure :: block_green => gpu_block_green
      procedure :: chebyshev_green => gpu_chebyshev_green
      procedure :: fetch_g0 => gpu_fetch_g0
   end type green_gpu

   interface green_gpu
      procedure :: gpu_constructor
   end interface green_gpu

contains

   !> Same construction as green.f90:101-113.
   function gpu_constructor(dos_obj) result(obj)
      type(green_gpu) :: obj
      type(dos), target, intent(in) :: dos_obj

      obj%dos => dos_obj
      obj%recursion => dos_obj%recursion
      obj%en => dos_obj%en
      obj%symbolic_atom => dos_obj%recursion%hamiltonian%charge%lattice%symbolic_atoms
      obj%lattice => dos_obj%recursion%lattice
      obj%control => dos_obj%recursion%lattice%control
      call obj%restore_to_default()
      if (rsrec_env_flag('RSREC_DEFER_G0')) obj%defer_g0 = .true.    ! (hosts that cannot reach the member: fortran/shadow/)
   end function gpu_constructor

   !> Replaces green.f90:1191-1339 (same interface; `g_out` is zeroed and the energy range ie_start .. ie_start+ie_len-1 filled).
   subroutine gpu_bgreen(this, g_out, i_site, ie_start, ie_len, a_inf, b_inf, eta)
      class(green_gpu), intent(inout) :: this
      integer, intent(in) :: i_site
      integer, intent(in) :: ie_start
      integer, intent(in) :: ie_len
      complex(rp), dimension(18, 18, this%en%channels_ldos + 10), intent(inout) :: g_out
      real(rp), dimension(18, 18), intent(in) :: a_inf
      real(rp), dimension(18, 18), intent(in) :: b_inf
      complex(rp), intent(in) :: eta
      !
      integer :: ll
      integer(c_int) :: rc, sym_i
      type(c_ptr) :: handle
      real(rp), allocatable, target :: ene(:), ai(:, :), bi(:, :)
      complex(rp), allocatable, target :: ab(:, :, :), bs(:, :, :), gt(:, :, :)

      g_out = (0.0d0, 0.0d0)
      if (ie_len <= 0) return
      ll = this%control%lld
      allocate (ene(ie_len), ai(18, 18), bi(18, 18), ab(18, 18, ll), bs(18, 18, ll), gt(18, 18, ie_len))
      ene = this%en%ene(ie_start:ie_start + ie_len - 1)
      ai = a_inf
      bi = b_inf
      ab = this%recursion%a_b(:, :, 1:ll, i_site)
      bs = this%recursion%b2_b(:, :, 1:ll, i_site)          ! sqrt(B^2): zsqr ran before (self.f90:829)
      sym_i = 0
      if (this%control%sym_term) sym_i = 1
      handle = rsrec_gpu_context()
      call g_timer%start('bgreen-gpu')
      rc = rsrec_block_green(handle, 1_c_int, int(ll, c_int), int(ie_len, c_int), c_loc(ene), real(eta, c_double), aimag(eta), sym_i, &
                             c_loc(ai), c_loc(bi), c_loc(ab), c_loc(bs), gt)
      call g_timer%stop('bgreen-gpu')
      if (rc /= 0) call g_logger%fatal('rsrec_block_green: '//rsrec_error_string(handle), __FILE__, __LINE__)
      g_out(:, :, ie_start:ie_start + ie_len - 1) = gt
   end subroutine gpu_bgreen

   !> Replaces green.f90:588-621: the per-site loop over `bgreen` becomes ONE library call for all sites of this rank, so the
   !> kernel of one chunk of sites overlaps the download of the previous one (rsrec.hip green_pipeline) instead of a
   !> launch + 13 MB copy per site.  The terminator comes from recursion%get_terminf (GPU kernel behind recursion_gpu).
   subroutine gpu_block_green(this)
      use mpi_mod, only: start_atom, end_atom, g2l_map, atoms_per_process
      class(green_gpu), intent(inout) :: this
      integer :: nw, ll, ldim, nv, nloc, n1, n, nw_site
      integer(c_int) :: rc, sym_i
      type(c_ptr) :: handle
      real(rp), dimension(this%lattice%nrec) :: a_inf0, b_inf0
      real(rp), dimension(18, 18, this%lattice%nrec) :: a_inf, b_inf
      real(rp), allocatable, target :: ene(:), ai(:, :, :), bi(:, :, :)
      complex(rp), allocatable, target :: ab(:, :, :, :), bs(:, :, :, :), gt(:, :, :, :)

      if (this%defer_g0 .and. .not. this%fetching) then
         this%g0_stale = .true.                               ! produced by fetch_g0 when somebody reads g0; repeated calls without a
         return                                               ! reader in between (block_green + calculate_fermi, DOS-only flows) stay free
      end if
      this%g0_stale = .false.
      ll = this%control%lld
      ldim = 18
      nw = 10*ll
      nloc = end_atom - start_atom + 1
      if (nloc <= 0) return
      n1 = g2l_map(start_atom)                               ! local indices of the rank's sites are contiguous (mpi.f90:72-78)
      ! Terminator: ONE get_terminf call for all sites of the rank (recursion.f90:2092).  With a recursion_gpu behind the class
      ! pointer this dispatches to the terminator kernel (one thread per site and matrix element, rsrec_terminator); with the
      ! reference's own type it is the reference's serial loop over the sites.
      a_inf = 0.0_rp; b_inf = 0.0_rp
      nw_site = nw
      call this%recursion%get_terminf(this%recursion%a_b(:, :, :, n1:n1 + nloc - 1), this%recursion%b2_b(:, :, :, n1:n1 + nloc - 1), nloc, &
                                      ll, ldim, nw_site, a_inf(:, :, n1:n1 + nloc - 1), b_inf(:, :, n1:n1 + nloc - 1), &
                                      a_inf0(n1:n1 + nloc - 1), b_inf0(n1:n1 + nloc - 1))
      nv = this%en%channels_ldos + 10
      allocate (ene(nv), ai(18, 18, nloc), bi(18, 18, nloc), ab(18, 18, ll, nloc), bs(18, 18, ll, nloc))
      ene = this%en%ene(1:nv)
      ai = a_inf(:, :, n1:n1 + nloc - 1)
      bi = b_inf(:, :, n1:n1 + nloc - 1)
      ab = this%recursion%a_b(:, :, 1:ll, n1:n1 + nloc - 1)
      bs = this%recursion%b2_b(:, :, 1:ll, n1:n1 + nloc - 1)  ! sqrt(B^2): zsqr ran before (self.f90:829)
      sym_i = 0
      if (this%control%sym_term) sym_i = 1
      handle = rsrec_gpu_context()
      call g_timer%start('bgreen-gpu')
      if (n1 == 1 .and. size(this%g0, 3) == nv .and. size(this%g0, 4) >= nloc) then
         ! green%g0(18,18,nv,atoms_per_process) (green.f90:192) is the library's output buffer itself: no staging copy of 13 MB per site
         rc = rsrec_block_green(handle, int(nloc, c_int), int(ll, c_int), int(nv, c_int), c_loc(ene), 0.0_c_double, 0.0_c_double, sym_i, &
                                c_loc(ai), c_loc(bi), c_loc(ab), c_loc(bs), this%g0)
      else
         allocate (gt(18, 18, nv, nloc))
         rc = rsrec_block_green(handle, int(nloc, c_int), int(ll, c_int), int(nv, c_int), c_loc(ene), 0.0_c_double, 0.0_c_double, sym_i, &
                                c_loc(ai), c_loc(bi), c_loc(ab), c_loc(bs), gt)
         if (rc == 0) this%g0(:, :, 1:nv, n1:n1 + nloc - 1) = gt
      end if
      call g_timer%stop('bgreen-gpu')
      if (rc /= 0) call g_logger%fatal('rsrec_block_green: '//rsrec_error_string(handle), __FILE__, __LINE__)
   end subroutine gpu_block_green

   !> `g0` of the last (deferred) `block_green` call, now.  A no-op when `g0` is up to date.
   subroutine gpu_fetch_g0(this)
      class(green_gpu), intent(inout) :: this
      if (.not. this%g0_stale) return
      this%fetching = .true.
      call this%block_green()                                ! the one call that does the work
      this%fetching = .false.
   end subroutine gpu_fetch_g0

   !> Replaces green.f90:1030-1108: g0 of the sites of this rank from the Chebyshev moments.  The side effect of the reference
   !> routine -- recursion%mu_ng = mu_n * Jackson kernel (* 2 beyond the first moment), read later by bands.f90:762 -- is kept.
   subroutine gpu_chebyshev_green(this)
      use mpi_mod, only: start_atom, end_atom, g2l_map
      use math_mod, only: jackson_kernel
      class(green_gpu), intent(inout) :: this
      integer :: n, n_glob, nv, nm, l, m, nloc, n1
      integer(c_int) :: rc
      type(c_ptr) :: handle
      real(rp), dimension(this%control%lld*2 + 2) :: kernel
      real(rp), allocatable, target :: ene(:)
      complex(rp), allocatable, target :: mu(:, :, :, :), gt(:, :, :, :)

      this%g0 = 0.0d0
      nv = this%en%channels_ldos + 10
      nm = this%control%lld*2 + 2
      nloc = end_atom - start_atom + 1
      if (nloc <= 0) return
      call jackson_kernel(nm, kernel)
      do n_glob = start_atom, end_atom
         n = g2l_map(n_glob)
         do l = 1, 18
            do m = 1, 18
               this%recursion%mu_ng(l, m, :, n) = this%recursion%mu_n(l, m, :, n)*kernel(:)
            end do
         end do
         this%recursion%mu_ng(:, :, 2:nm, n) = this%recursion%mu_ng(:, :, 2:nm, n)*2.0_rp
      end do
      n1 = g2l_map(start_atom)
      allocate (ene(nv), mu(18, 18, nm, nloc))
      ene = this%en%ene(1:nv)
      mu = this%recursion%mu_n(:, :, 1:nm, n1:n1 + nloc - 1)
      handle = rsrec_gpu_context()
      call g_timer%start('chebyshev-green-gpu')
      if (n1 == 1 .and. size(this%g0, 3) == nv .and. size(this%g0, 4) >= nloc) then
         rc = rsrec_chebyshev_green(handle, int(nloc, c_int), int(this%control%lld, c_int), int(nv, c_int), c_loc(ene), &
                                    real(this%en%energy_min, c_double), real(this%en%energy_max, c_double), c_loc(mu), this%g0)
      else
         allocate (gt(18, 18, nv, nloc))
         rc = rsrec_chebyshev_green(handle, int(nloc, c_int), int(this%control%lld, c_int), int(nv, c_int), c_loc(ene), &
                                    real(this%en%energy_min, c_double), real(this%en%energy_max, c_double), c_loc(mu), gt)
         if (rc == 0) this%g0(:, :, 1:nv, n1:n1 + nloc - 1) = gt
      end if
      call g_timer%stop('chebyshev-green-gpu')
      if (rc /= 0) call g_logger%fatal('rsrec_chebyshev_green: '//rsrec_error_string(handle), __FILE__, __LINE__)
   end subroutine gpu_chebyshev_green

end module green_gpu_mod
