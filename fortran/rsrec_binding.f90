!------------------------------------------------------------------------------
! rsrec_binding -- iso_c_binding interface of librsrec (include/rsrec.h).
!
! The reference is 100 % Fortran and has no FFI today; these are the bind(C)
! interfaces the GPU recursion type (recursion_gpu.f90) calls.  Every array is
! passed exactly as the reference stores it (column-major, complex(rp) =
! interleaved re/im, 1-based atom numbers), so no host-side reshuffling.
!------------------------------------------------------------------------------
module rsrec_binding
   use, intrinsic :: iso_c_binding
   implicit none
   public

   interface
      function rsrec_version() bind(C, name='rsrec_version') result(v)
         import :: c_int
         integer(c_int) :: v
      end function

      function rsrec_device_count() bind(C, name='rsrec_device_count') result(n)
         import :: c_int
         integer(c_int) :: n
      end function

      function rsrec_create(handle, device) bind(C, name='rsrec_create') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), intent(out) :: handle
         integer(c_int), value :: device
         integer(c_int) :: rc
      end function

      function rsrec_destroy(handle) bind(C, name='rsrec_destroy') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int) :: rc
      end function

      function rsrec_set_lattice(handle, kk, nncols, nn, iz, nmax, ntype) bind(C, name='rsrec_set_lattice') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: kk, nncols, nmax, ntype
         type(c_ptr), value :: nn, iz
         integer(c_int) :: rc
      end function

      function rsrec_block_green(handle, nsites, lld, nen, ene, eta_re, eta_im, sym_term, a_inf, b_inf, a_b, b_sqrt, g0) &
         bind(C, name='rsrec_block_green') result(rc)
         import :: c_int, c_ptr, c_double, c_double_complex
         type(c_ptr), value :: handle
         integer(c_int), value :: nsites, lld, nen, sym_term
         real(c_double), value :: eta_re, eta_im
         type(c_ptr), value :: ene, a_inf, b_inf, a_b, b_sqrt
         complex(c_double_complex), dimension(*), intent(inout) :: g0   ! by address: green%g0 itself can be the target (833 MB for 64 sites: no staging copy)
         integer(c_int) :: rc
      end function

      function rsrec_chebyshev_green(handle, nsites, lld, nen, ene, energy_min, energy_max, mu_n, g0) &
         bind(C, name='rsrec_chebyshev_green') result(rc)
         import :: c_int, c_ptr, c_double, c_double_complex
         type(c_ptr), value :: handle
         integer(c_int), value :: nsites, lld, nen
         real(c_double), value :: energy_min, energy_max
         type(c_ptr), value :: ene, mu_n
         complex(c_double_complex), dimension(*), intent(inout) :: g0   ! by address (see rsrec_block_green)
         integer(c_int) :: rc
      end function

      function rsrec_set_positions(handle, cr) bind(C, name='rsrec_set_positions') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         type(c_ptr), value :: cr
         integer(c_int) :: rc
      end function

      function rsrec_set_hamiltonian(handle, nslots, hoh, nsp, ee, lsham, eeo, enim, hall, hallo) &
         bind(C, name='rsrec_set_hamiltonian') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: nslots, hoh, nsp
         type(c_ptr), value :: ee, lsham, eeo, enim, hall, hallo
         integer(c_int) :: rc
      end function

      function rsrec_assemble_blocks(handle, part, ncls, nslots, hoh, hmag, nbr_type, obarm, ntype, blocks, blocks_o) &
         bind(C, name='rsrec_assemble_blocks') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: part, ncls, nslots, hoh, ntype
         type(c_ptr), value :: hmag, nbr_type, obarm, blocks, blocks_o
         integer(c_int) :: rc
      end function

      function rsrec_block_lanczos(handle, nsites, seed_atoms, lld, a_b, b2_b) bind(C, name='rsrec_block_lanczos') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: nsites, lld
         type(c_ptr), value :: seed_atoms, a_b, b2_b
         integer(c_int) :: rc
      end function

      function rsrec_block_lanczos_seeded(handle, nchains, nseed, seed_atoms, seed_coef, lld, a_b, b2_b) &
         bind(C, name='rsrec_block_lanczos_seeded') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: nchains, nseed, lld
         type(c_ptr), value :: seed_atoms, seed_coef, a_b, b2_b
         integer(c_int) :: rc
      end function

      function rsrec_block_lanczos_local_axis(handle, nsites, seed_atoms, rot, lld, a_b, b2_b) &
         bind(C, name='rsrec_block_lanczos_local_axis') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: nsites, lld
         type(c_ptr), value :: seed_atoms, rot, a_b, b2_b
         integer(c_int) :: rc
      end function

      function rsrec_pack_diag(handle, site_offset, nsites_total, a_img, b2_img) bind(C, name='rsrec_pack_diag') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: site_offset, nsites_total
         type(c_ptr), value :: a_img, b2_img
         integer(c_int) :: rc
      end function

      function rsrec_terminator(handle, nsites, lld, a_b, b_sqrt, a_inf, b_inf, a_inf0, b_inf0) bind(C, name='rsrec_terminator') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: nsites, lld
         type(c_ptr), value :: a_b, b_sqrt, a_inf, b_inf, a_inf0, b_inf0
         integer(c_int) :: rc
      end function

      function rsrec_scalar_density(handle, nsites, nmdir, llmax, lld, a, b2, npts, ene, dw_l, cshi, tdens) bind(C, name='rsrec_scalar_density') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: nsites, nmdir, llmax, lld, npts
         type(c_ptr), value :: a, b2, ene, dw_l, cshi, tdens
         integer(c_int) :: rc
      end function

      function rsrec_block_ldos(handle, nen, ene, eta_re, eta_im, sym_term, site_offset, nsites_total, dtot, dosia, dosial, a_inf, b_inf) &
         bind(C, name='rsrec_block_ldos') result(rc)
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: handle
         integer(c_int), value :: nen, sym_term, site_offset, nsites_total
         real(c_double), value :: eta_re, eta_im
         type(c_ptr), value :: ene, dtot, dosia, dosial, a_inf, b_inf
         integer(c_int) :: rc
      end function

      function rsrec_kubo_moments(handle, nvec, nseed, seed_atoms, seed_coef, cond_ll, a, b, v_a, vo_a, v_b, vo_b, mu_nm) &
         bind(C, name='rsrec_kubo_moments') result(rc)
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: handle
         integer(c_int), value :: nvec, nseed, cond_ll
         real(c_double), value :: a, b
         type(c_ptr), value :: seed_atoms, seed_coef, v_a, vo_a, v_b, vo_b, mu_nm
         integer(c_int) :: rc
      end function

      function rsrec_apply_operator(handle, vel, v_op, vo_op, psi_in, psi_out, a, b) bind(C, name='rsrec_apply_operator') result(rc)
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: handle
         integer(c_int), value :: vel
         real(c_double), value :: a, b
         type(c_ptr), value :: v_op, vo_op, psi_in, psi_out
         integer(c_int) :: rc
      end function

      function rsrec_zsqr(handle, nmat, b2_b) bind(C, name='rsrec_zsqr') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: nmat
         type(c_ptr), value :: b2_b
         integer(c_int) :: rc
      end function

      function rsrec_chebyshev(handle, nsites, seed_atoms, lld, a, b, mu_n) bind(C, name='rsrec_chebyshev') result(rc)
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: handle
         integer(c_int), value :: nsites, lld
         real(c_double), value :: a, b
         type(c_ptr), value :: seed_atoms, mu_n
         integer(c_int) :: rc
      end function

      function rsrec_chebyshev_seeded(handle, nchains, nseed, seed_atoms, seed_coef, lld, a, b, mu_n) &
         bind(C, name='rsrec_chebyshev_seeded') result(rc)
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: handle
         integer(c_int), value :: nchains, nseed, lld
         real(c_double), value :: a, b
         type(c_ptr), value :: seed_atoms, seed_coef, mu_n
         integer(c_int) :: rc
      end function

      function rsrec_scalar_lanczos(handle, nsites, seed_atoms, lld, llmax, a, b2) bind(C, name='rsrec_scalar_lanczos') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: nsites, lld, llmax
         type(c_ptr), value :: seed_atoms, a, b2
         integer(c_int) :: rc
      end function

      subroutine rsrec_site_partition(rank, nprocs, nsites, start_atom, end_atom) bind(C, name='rsrec_site_partition')
         import :: c_int
         integer(c_int), value :: rank, nprocs, nsites
         integer(c_int), intent(out) :: start_atom, end_atom
      end subroutine

      function rsrec_orbital_moments(handle, nseeds, seed_atoms, lld, a, b, cr, alat, mu_orb, mu_seed) bind(C, name='rsrec_orbital_moments') result(rc)
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: handle
         integer(c_int), value :: nseeds, lld
         real(c_double), value :: a, b, alat
         type(c_ptr), value :: seed_atoms, cr, mu_orb, mu_seed
         integer(c_int) :: rc
      end function

      function rsrec_pack_moments(handle, site_offset, nsites_total, mu_img) bind(C, name='rsrec_pack_moments') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), value :: site_offset, nsites_total
         type(c_ptr), value :: mu_img
         integer(c_int) :: rc
      end function

      ! library-level communicator (RCCL inside librsrec; include/rsrec.h): the MPI_ALLREDUCE(MPI_IN_PLACE, ..., MPI_SUM) of
      ! bands.f90:271-274 without MPI on the host
      function rsrec_comm_unique_id(id) bind(C, name='rsrec_comm_unique_id') result(rc)
         import :: c_int, c_char
         character(kind=c_char), intent(out) :: id(*)          ! 128 bytes
         integer(c_int) :: rc
      end function

      function rsrec_comm_init(handle, rank, nranks, id) bind(C, name='rsrec_comm_init') result(rc)
         import :: c_int, c_ptr, c_char
         type(c_ptr), value :: handle
         integer(c_int), value :: rank, nranks
         character(kind=c_char), intent(in) :: id(*)
         integer(c_int) :: rc
      end function

      function rsrec_comm_init_file(handle, rank, nranks, path, timeout_s) bind(C, name='rsrec_comm_init_file') result(rc)
         import :: c_int, c_ptr, c_char, c_double
         type(c_ptr), value :: handle
         integer(c_int), value :: rank, nranks
         character(kind=c_char), intent(in) :: path(*)         ! NUL-terminated
         real(c_double), value :: timeout_s
         integer(c_int) :: rc
      end function

      function rsrec_allreduce_sum(handle, buf, n) bind(C, name='rsrec_allreduce_sum') result(rc)
         import :: c_int, c_ptr, c_size_t
         type(c_ptr), value :: handle, buf
         integer(c_size_t), value :: n
         integer(c_int) :: rc
      end function

      function rsrec_comm_size(handle, rank, nranks) bind(C, name='rsrec_comm_size') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int), intent(out) :: rank, nranks
         integer(c_int) :: rc
      end function

      function rsrec_comm_destroy(handle) bind(C, name='rsrec_comm_destroy') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: handle
         integer(c_int) :: rc
      end function

      function rsrec_last_error(handle, buf, n) bind(C, name='rsrec_last_error') result(rc)
         import :: c_int, c_ptr, c_char, c_size_t
         type(c_ptr), value :: handle
         character(kind=c_char), intent(out) :: buf(*)
         integer(c_size_t), value :: n
         integer(c_int) :: rc
      end function

      function rsrec_set_option(handle, key, val) bind(C, name='rsrec_set_option') result(rc)
         import :: c_int, c_ptr, c_char, c_long
         type(c_ptr), value :: handle
         character(kind=c_char), intent(in) :: key(*)
         integer(c_long), value :: val
         integer(c_int) :: rc
      end function

      function rsrec_get_timing(handle, out, n) bind(C, name='rsrec_get_timing') result(rc)
         import :: c_int, c_ptr, c_double
         type(c_ptr), value :: handle
         real(c_double), intent(out) :: out(*)
         integer(c_int), value :: n
         integer(c_int) :: rc
      end function
   end interface

contains

   !> Error text of a handle as a Fortran string
   function rsrec_error_string(handle) result(msg)
      type(c_ptr), intent(in) :: handle
      character(len=:), allocatable :: msg
      character(kind=c_char) :: buf(512)
      integer :: i, rc
      rc = rsrec_last_error(handle, buf, int(512, c_size_t))
      msg = ''
      do i = 1, 512
         if (buf(i) == c_null_char) exit
         msg = msg//buf(i)
      end do
   end function rsrec_error_string

end module rsrec_binding
