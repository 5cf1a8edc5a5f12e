!------------------------------------------------------------------------------
! rsrec_context_mod -- the per-process device context of the GPU drop-in types.
!
! ONE module-level handle per process, created lazily on the first call that needs the device.  Nothing device-side lives in
! any of the drop-in types: every caller in the reference constructs its objects by intrinsic assignment from a function
! result (calculation.f90:270, :599 ...) and the reference's types have `final` procedures (recursion.f90:115,150) -- a handle
! stored in a type would be shallow-copied and then destroyed with the temporary (SURVEY.md 8b, "finalizer trap").
! MPI rank r uses GPU mod(r, device_count) (mpi_mod's `rank`, set by main.f90:47 / get_mpi_variables).
!
! A module of its own (it was part of recursion_gpu_mod until round 4) so that hamiltonian_gpu_mod, which the recursion's own
! module graph depends on once `hamiltonian_mod` is shadowed (fortran/shadow/), can reach the context without a cycle.
!------------------------------------------------------------------------------
module rsrec_context_mod
   use, intrinsic :: iso_c_binding
   use mpi_mod, only: rank
   use logger_mod, only: g_logger
   use rsrec_binding
   implicit none
   private
   public :: rsrec_gpu_context, rsrec_gpu_shutdown, rsrec_env_flag, g_handle

   !> the per-process device context (lazy)
   type(c_ptr), save :: g_handle = c_null_ptr

contains

   !> The per-process device context, created on first use.
   function rsrec_gpu_context() result(handle)
      type(c_ptr) :: handle
      integer(c_int) :: rc, ndev
      if (.not. c_associated(g_handle)) then
         ndev = rsrec_device_count()
         if (ndev <= 0) call g_logger%fatal('rsrec: no usable GPU (librsrec has no CPU fallback)', __FILE__, __LINE__)
         rc = rsrec_create(g_handle, int(mod(rank, ndev), c_int))
         if (rc /= 0) call g_logger%fatal('rsrec: rsrec_create failed', __FILE__, __LINE__)
      end if
      handle = g_handle
   end function rsrec_gpu_context

   !> Release the device context (optional; call once before MPI_FINALIZE).
   subroutine rsrec_gpu_shutdown()
      integer(c_int) :: rc
      if (c_associated(g_handle)) rc = rsrec_destroy(g_handle)
      g_handle = c_null_ptr
   end subroutine rsrec_gpu_shutdown

   !> .true. if the environment variable `name` is set to a non-empty value: run-time switches of the drop-in types for hosts that
   !> cannot set their members (the reference's unmodified calculation.f90 behind the shadow modules): RSREC_HOST_LDOS, RSREC_HOST_HAM,
   !> RSREC_DEFER_G0
   function rsrec_env_flag(name) result(set)
      character(len=*), intent(in) :: name
      logical :: set
      character(len=8) :: val
      integer :: n, stat
      call get_environment_variable(name, val, n, stat)
      set = (stat == 0 .or. stat == -1) .and. n > 0
   end function rsrec_env_flag
end module rsrec_context_mod
