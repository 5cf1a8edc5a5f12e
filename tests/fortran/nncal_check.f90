!------------------------------------------------------------------------------
! nncal_check -- runs the reference's lattice pre-processing (build_data, bravais, [build_surf_full], [newclu], structb)
! with either the reference's own `type(lattice)` (argument `ref`) or `type(lattice_cells)` (argument `cells`) and dumps
! the resulting neighbour table lattice%nn to nn_<mode>.bin (int32 stream: kk, ncol, nn(kk, ncol)); the `map` and
! `clust` files the run leaves behind are the reference's wire formats.  tests/test_lattice_cells.py runs both modes in two
! scratch copies of a case directory and compares the tables and files byte for byte.  Reads input.nml.
!------------------------------------------------------------------------------
program nncal_check
   use mpi_mod
   use control_mod
   use lattice_mod
   use lattice_cells_mod
   use calculation_mod
   use timer_mod, only: g_timer, timer
   implicit none
   type(calculation) :: calc_obj
   type(control), target :: control_obj
   class(lattice), pointer :: lat
   character(len=32) :: mode, pre
   integer(8) :: c0, c1, rate
   integer :: u

   rank = 0
   numprocs = 1
   g_timer = timer()
   call get_command_argument(1, mode)
   calc_obj = calculation('input.nml')
   pre = trim(calc_obj%pre_processing)
   control_obj = control('input.nml')
   if (trim(mode) == 'cells') then
      allocate (lattice_cells :: lat)
   else
      allocate (lattice :: lat)
   end if
   select type (lat)
   type is (lattice)
      lat = lattice(control_obj)
   type is (lattice_cells)
      lat%lattice = lattice(control_obj)
   end select
   call lat%build_data()
   call lat%bravais()
   call system_clock(c0, rate)
   select case (trim(pre))
   case ('bravais')
      call lat%structb(.false.)
   case ('buildsurf')
      call lat%build_surf_full()
      call lat%structb(.false.)
   case ('newclubulk')
      call lat%newclu()
      call lat%structb(.false.)
   case ('newclusurf')
      call lat%build_surf_full()
      call lat%newclu()
      call lat%structb(.false.)
   case default
      stop 'nncal_check: unsupported pre_processing'
   end select
   call system_clock(c1)
   open (newunit=u, file='nn_'//trim(mode)//'.bin', access='stream', form='unformatted', status='replace')
   write (u) int(lat%kk, 4), int(size(lat%nn, 2), 4)
   write (u) int(lat%nn, 4)
   close (u)
   write (*, '(a, a, a, i8, a, i5, a, f10.3, a)') 'nncal_check ', trim(mode), ': kk =', lat%kk, ', columns =', size(lat%nn, 2), &
      ', neighbour search + structb ', real(c1 - c0)/real(rate), ' s'
   select type (lat)
   type is (lattice_cells)
      write (*, '(a, i14, a, i14)') 'pairs evaluated ', lat%pairs_evaluated, ' of ', int(lat%kk, 8)*(lat%kk - 1)/2
   end select
end program nncal_check
