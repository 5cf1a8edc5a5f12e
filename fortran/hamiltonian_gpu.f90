! Host side of rsrec, fourth drop-in type: the raw operator blocks assembled on the GPU (SURVEY 8 f2, second half).
!
! `type, extends(hamiltonian) :: hamiltonian_gpu` overrides TWO bindings, build_bulkham (hamiltonian.f90:1553-1616) and build_locham
! (:1618-1667).  Everything that produces their inputs stays the reference's own code and runs on the host exactly as before:
! chbar_nc (:2305: structure constants x potential parameters -> the four 9x9 parts Hx, Hy, Hz, H0 in `hmag`), build_obarm (:1486),
! build_enim (:1509).  What moves to the device is what the two routines do with those inputs for every class atom and slot m:
!     ee(:,:,m)  = [[H0 + Hz, Hx - i Hy], [Hx + i Hy, H0 - Hz]]                   (:1565-1570 / :1631-1636)
!     eeo(:,:,m) = ee(:,:,m) * obarm(:,:,type of the atom behind slot m)            (:1599 / :1654, zgemm)
! in ONE call per routine (rsrec_assemble_blocks, include/rsrec.h).  The results come back into this%ee / %eeo / %hall / %hallo, because
! the reference's host routines read them too (block_to_sparse, the exchange, rotate_to_local_axis), and they STAY on the device: when
! recursion_gpu then hands the same arrays to rsrec_set_hamiltonian they are recognised bit for bit and not uploaded again.
!
! Use (one declaration and one construction line in the caller, INTEGRATION.md section 3c):
!     type(hamiltonian_gpu), target :: hamiltonian_obj ;  hamiltonian_obj%hamiltonian = hamiltonian(charge_obj)
!
! Not reproduced: the reference's debug dumps of ee to units 131 / 132 (:1578-1581) and the unused product eeoee = eeo * ee^H is formed on
! the host from the downloaded blocks (:1600; nothing in the reference reads it).
module hamiltonian_gpu_mod
   use, intrinsic :: iso_c_binding
   use hamiltonian_mod
   use charge_mod, only: charge
   use precision_mod, only: rp
   use math_mod, only: i_unit, cone, czero
   use logger_mod, only: g_logger
   use timer_mod, only: g_timer
   use rsrec_context_mod, only: rsrec_gpu_context, rsrec_env_flag
   use rsrec_binding
   implicit none
   private

   type, public, extends(hamiltonian) :: hamiltonian_gpu
      !> .false.: both routines run the reference's host code (the inherited bindings)
      logical :: device_assembly = .true.
      !> calls that went through the device (diagnostic)
      integer :: n_device_assemblies = 0
   contains
      procedure :: build_bulkham => gpu_build_bulkham
      procedure :: build_locham => gpu_build_locham
   end type hamiltonian_gpu

   interface hamiltonian_gpu
      procedure :: gpu_constructor
   end interface hamiltonian_gpu

contains

   !> Same construction as hamiltonian.f90:129-139 (pointers, restore_to_default, build_from_file), on the extended object itself.
   !> RSREC_HOST_HAM set in the environment: device_assembly = .false. (the switch for hosts that cannot reach the member -- the
   !> reference's unmodified calculation.f90 behind fortran/shadow/hamiltonian_mod.f90).
   function gpu_constructor(charge_obj) result(obj)
      type(hamiltonian_gpu) :: obj
      type(charge), target, intent(in) :: charge_obj

      obj%charge => charge_obj
      obj%lattice => charge_obj%lattice
      obj%control => charge_obj%lattice%control
      call obj%restore_to_default()
      call obj%build_from_file()
      if (rsrec_env_flag('RSREC_HOST_HAM')) obj%device_assembly = .false.
   end function gpu_constructor

   !> Collect what chbar_nc left for class atom `ia` into column `icls` of the device call's inputs; the neighbour types as :1586-1597.
   subroutine collect_class(this, ia, icls, nr, hm, ty)
      class(hamiltonian_gpu), intent(inout) :: this
      integer, intent(in) :: ia, icls, nr
      complex(rp), intent(inout) :: hm(:, :, :, :, :)
      integer(c_int), intent(inout) :: ty(:, :)
      integer :: m, ja
      hm(:, :, 1:nr, :, icls) = this%hmag(:, :, 1:nr, :)
      do m = 1, nr
         if (m > 1) then
            ja = this%charge%lattice%nn(ia, m)
            if (ja /= 0) ty(m, icls) = int(this%charge%lattice%iz(ja), c_int)
         else
            ty(m, icls) = int(this%charge%lattice%iz(ia), c_int)
         end if
      end do
   end subroutine collect_class

   subroutine assemble(this, part, ncls, hm, ty, obarm, blocks, blocks_o)
      class(hamiltonian_gpu), intent(inout) :: this
      integer, intent(in) :: part, ncls
      complex(rp), intent(in), target, contiguous :: hm(:, :, :, :, :), obarm(:, :, :)
      integer(c_int), intent(in), target, contiguous :: ty(:, :)
      complex(rp), intent(inout), target, contiguous :: blocks(:, :, :, :), blocks_o(:, :, :, :)
      integer(c_int) :: rc, hoh_i
      type(c_ptr) :: ctx
      hoh_i = 0
      if (this%hoh) hoh_i = 1
      ctx = rsrec_gpu_context()
      rc = rsrec_assemble_blocks(ctx, int(part, c_int), int(ncls, c_int), int(size(hm, 3), c_int), hoh_i, c_loc(hm), c_loc(ty), c_loc(obarm), &
                                 int(size(obarm, 3), c_int), c_loc(blocks), c_loc(blocks_o))
      if (rc /= 0) call g_logger%fatal('hamiltonian_gpu: rsrec_assemble_blocks: '//rsrec_error_string(ctx), __FILE__, __LINE__)
      this%n_device_assemblies = this%n_device_assemblies + 1
   end subroutine assemble

   subroutine gpu_build_bulkham(this)
      class(hamiltonian_gpu), intent(inout) :: this
      complex(rp), allocatable :: hm(:, :, :, :, :)
      integer(c_int), allocatable :: ty(:, :)
      integer :: ntype, it, ia, ino, nr, m, i, j, nsl

      if (.not. this%device_assembly) then
         call this%hamiltonian%build_bulkham()
         return
      end if
      ntype = this%charge%lattice%ntype
      nsl = size(this%ee, 3)
      allocate (hm(9, 9, nsl, 4, ntype), ty(nsl, ntype))
      hm = (0.0_rp, 0.0_rp)
      ty = 0_c_int
      do it = 1, ntype
         ia = this%charge%lattice%atlist(it)
         ino = this%charge%lattice%num(ia)
         nr = this%charge%lattice%nn(ia, 1)
         call this%chbar_nc(ia, nr, ino, it)
         call collect_class(this, ia, it, nr, hm, ty)
         do m = 1, nr                                   ! the magnetic part alone (:1572-1576), host as before
            do i = 1, 9
               do j = 1, 9
                  this%hxc(j, i, m, it) = this%hmag(j, i, m, 3)
                  this%hxc(j + 9, i + 9, m, it) = -this%hmag(j, i, m, 3)
                  this%hxc(j, i + 9, m, it) = this%hmag(j, i, m, 1) - i_unit*this%hmag(j, i, m, 2)
                  this%hxc(j + 9, i, m, it) = this%hmag(j, i, m, 1) + i_unit*this%hmag(j, i, m, 2)
               end do
            end do
         end do
      end do
      if (this%hoh) then
         call this%build_obarm()
         call this%build_enim()
      end if
      call assemble(this, 0, ntype, hm, ty, this%obarm, this%ee, this%eeo)
      if (this%hoh) then
         do it = 1, ntype
            do m = 1, this%charge%lattice%nn(this%charge%lattice%atlist(it), 1)
               if (ty(m, it) > 0) call zgemm('n', 'c', 18, 18, 18, cone, this%eeo(:, :, m, it), 18, this%ee(:, :, m, it), 18, czero, this%eeoee(:, :, m, it), 18)
            end do
         end do
      end if
      if (this%local_axis) then
         this%ee_glob = this%ee
         if (this%hoh) this%eeo_glob = this%eeo
      end if
   end subroutine gpu_build_bulkham

   subroutine gpu_build_locham(this)
      class(hamiltonian_gpu), intent(inout) :: this
      complex(rp), allocatable :: hm(:, :, :, :, :)
      integer(c_int), allocatable :: ty(:, :)
      integer :: nmax, nlim, ino, nr, nsl

      if (.not. this%device_assembly) then
         call this%hamiltonian%build_locham()
         return
      end if
      call g_timer%start('build local hamiltonian')
      nmax = this%charge%lattice%nmax
      nsl = size(this%hall, 3)
      allocate (hm(9, 9, nsl, 4, nmax), ty(nsl, nmax))
      hm = (0.0_rp, 0.0_rp)
      ty = 0_c_int
      do nlim = 1, nmax
         nr = this%charge%lattice%nn(nlim, 1)
         ino = this%charge%lattice%num(nlim)
         call this%chbar_nc(nlim, nr, ino, nlim)
         call collect_class(this, nlim, nlim, nr, hm, ty)
      end do
      if (this%hoh) then
         call this%build_obarm()
         call this%build_enim()
      end if
      call assemble(this, 1, nmax, hm, ty, this%obarm, this%hall, this%hallo)
      if (this%local_axis) then
         this%hall_glob = this%hall
         if (this%hoh) this%hallo_glob = this%hallo
      end if
      call g_timer%stop('build local hamiltonian')
   end subroutine gpu_build_locham

end module hamiltonian_gpu_mod
