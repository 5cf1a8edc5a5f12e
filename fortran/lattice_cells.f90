!------------------------------------------------------------------------------
! lattice_cells -- O(N) neighbour search behind the reference's own lattice type (SURVEY.md 8 f3).
!
! `type, extends(lattice) :: lattice_cells` overrides ONE binding, `nncal` (lattice.f90:3035-3118), the all-pairs
! search behind `structb` (:1851) and `newclu` (:1682-1730): kk^2/2 pair distances, each through the 27 periodic images of
! `f_wrap_coord_diff` (:2965) when `pbc` is set -- 74 s for the 10 648-atom cell of BASELINE config 1, hours at 10^5 atoms,
! against 0.5 s for the GPU recursion of 64 sites on it.  Everything else (`structb`, `remd`, the `map` / `clust` files,
! `outmap`, `dbar1`) is inherited and sees the same table.
!
! The override bins the atoms into cells at least one cut-off wide (in the fractional coordinates of the periodic box,
! so that skewed cells work) and evaluates only pairs from adjacent cells -- with the reference's own expressions:
! the same `f_wrap_coord_diff` call, the same `sum(d**2)`, the same NGBR decision function, so a pair is a neighbour here
! exactly when it is one there.  Candidate generation only has to be a superset: a pair the reference accepts has an image
! closer than the cut-off, hence fractional separations below one cell width in every direction.  The reference's lists are
! in ascending atom order (pairs (I, J < I) are visited with I ascending, J ascending, and appended to both atoms); here the
! lists are sorted at the end.  The partial-update mode (negative IZP entries, :3070-3083) is passed on to the parent.
!------------------------------------------------------------------------------
module lattice_cells_mod
   use precision_mod, only: rp
   use lattice_mod
   use control_mod, only: control
   implicit none
   private
   public :: lattice_cells

   type, extends(lattice) :: lattice_cells
      !> pairs whose distance was evaluated by the last nncal call (the reference evaluates nat*(nat-1)/2)
      integer(8) :: pairs_evaluated = 0_8
   contains
      procedure :: nncal => nncal_cells
   end type lattice_cells

   interface lattice_cells
      procedure :: cells_constructor
   end interface lattice_cells

contains

   !> Same construction as lattice.f90:361-370 (control pointer, restore_to_default, build_from_file), on the extended object itself.
   function cells_constructor(control_obj) result(obj)
      type(lattice_cells) :: obj
      type(control), target, intent(in) :: control_obj

      obj%control => control_obj
      call obj%restore_to_default()
      call obj%build_from_file()
   end function cells_constructor

   subroutine nncal_cells(this, ct, crd, ndim, nat, izp, nn, nd, nm, ngbr, ntot)
      class(lattice_cells), intent(inout) :: this
      integer, intent(in) :: nat, nd, ndim, ntot
      integer, dimension(nat), intent(in) :: izp
      real(rp), dimension(ndim, nat), intent(in) :: crd
      integer, intent(inout) :: nm
      integer, dimension(nd, nm), intent(inout) :: nn
      real(rp), dimension(50), intent(inout) :: ct
      integer, external :: ngbr

      real(rp) :: box(3, 3), rec(3, 3), det, w(3), s(3), smin(3), smax(3), rc, r2, ddum(3), dum(1)
      logical :: per(3)
      integer :: nc(3), c(3), cj(3), d1, d2, d3, nv(3), lo(3)
      integer, allocatable :: cell_of(:, :), head(:), next(:)
      integer :: i, j, l, id, icell, nnmax, ncol, iip, jjp, k, t
      integer(8) :: ncell_tot

      rc = ct(1)                                      ! mapa, the decision function every caller passes, compares r2 with ct(1)**2 (:2937-2951)
      if (any(izp(1:nat) < 0) .or. ndim /= 3 .or. .not. (rc > 0.0_rp) .or. nat < 64) then
         call this%lattice%nncal(ct, crd, ndim, nat, izp, nn, nd, nm, ngbr, ntot)
         return
      end if

      ! periodic box of f_wrap_coord_diff (:2993-2996) and its reciprocal rows; Cartesian cells without pbc
      per = .false.
      box = 0.0_rp
      do l = 1, 3
         box(l, l) = 1.0_rp
      end do
      if (this%pbc) then
         per = [this%b1, this%b2, this%b3]
         box(:, 1) = (this%n1)*this%a(:, 1)*this%alat
         box(:, 2) = (this%n2)*this%a(:, 2)*this%alat
         box(:, 3) = (this%n3)*this%a(:, 3)*this%alat
      end if
      rec(1, :) = cross(box(:, 2), box(:, 3))
      rec(2, :) = cross(box(:, 3), box(:, 1))
      rec(3, :) = cross(box(:, 1), box(:, 2))
      det = dot_product(box(:, 1), rec(1, :))
      if (abs(det) < 1.0e-12_rp*maxval(abs(box))**3) then
         call this%lattice%nncal(ct, crd, ndim, nat, izp, nn, nd, nm, ngbr, ntot)
         return
      end if
      rec = rec/det
      ! |delta s_l| = |rec(l,:) . delta r| <= |rec(l,:)| |delta r|: cells at least this wide (plus a rounding margin) in direction l
      do l = 1, 3
         w(l) = norm2(rec(l, :))*rc*(1.0_rp + 1.0e-9_rp)
      end do

      allocate (cell_of(3, nat))
      smin = huge(1.0_rp); smax = -huge(1.0_rp)
      do i = 1, nat
         s = matmul(rec, crd(:, i))
         smin = min(smin, s); smax = max(smax, s)
      end do
      do l = 1, 3
         if (per(l)) then
            nc(l) = max(1, int(1.0_rp/w(l)))
         else
            nc(l) = max(1, int((smax(l) - smin(l))/w(l)))
         end if
      end do
      nc = min(nc, 1024)                              ! (wider cells stay correct; keeps the products below within integer(8))
      ! no more cells than atoms are worth having (merging cells keeps them wide enough)
      do while (int(nc(1), 8)*nc(2)*nc(3) > 8_8*nat)
         l = maxloc(nc, 1)
         nc(l) = (nc(l) + 1)/2
      end do
      do i = 1, nat
         s = matmul(rec, crd(:, i))
         do l = 1, 3
            if (per(l)) then
               cell_of(l, i) = min(nc(l) - 1, int((s(l) - floor(s(l)))*nc(l)))
            else
               cell_of(l, i) = min(nc(l) - 1, int((s(l) - smin(l))/(smax(l) - smin(l) + tiny(1.0_rp))*nc(l)))
            end if
         end do
      end do
      ncell_tot = int(nc(1), 8)*nc(2)*nc(3)
      allocate (head(ncell_tot), next(nat))
      head = 0

      nn(1:nat, 1) = 1
      ncol = 1                                        ! columns 2 .. ncol of nn are initialised (the reference zeroes all nm of them)
      nnmax = 0
      this%pairs_evaluated = 0_8
      do l = 1, 3                                     ! adjacent cells per direction: -1, 0, +1, without visiting a cell twice
         if (per(l)) then
            nv(l) = min(3, nc(l)); lo(l) = merge(-1, 0, nc(l) >= 3)
         else
            nv(l) = 3; lo(l) = -1
         end if
      end do
      ! atoms enter the cell lists as they are visited, so a cell holds exactly the atoms J < I
      do i = 1, nat
         c = cell_of(:, i)
         iip = abs(izp(i))
         do d3 = lo(3), lo(3) + nv(3) - 1
            cj(3) = neighbour_cell(c(3), d3, nc(3), per(3)); if (cj(3) < 0) cycle
            do d2 = lo(2), lo(2) + nv(2) - 1
               cj(2) = neighbour_cell(c(2), d2, nc(2), per(2)); if (cj(2) < 0) cycle
               do d1 = lo(1), lo(1) + nv(1) - 1
                  cj(1) = neighbour_cell(c(1), d1, nc(1), per(1)); if (cj(1) < 0) cycle
                  j = head(1 + cj(1) + nc(1)*(cj(2) + nc(2)*cj(3)))
                  do while (j /= 0)
                     jjp = abs(izp(j))
                     r2 = 0.0
                     if (this%pbc) then
                        call this%f_wrap_coord_diff(nat, crd, i, j, ddum)
                        r2 = sum(ddum(:)**2)
                     else
                        do l = 1, 3
                           ddum(l) = crd(l, i) - crd(l, j)
                           r2 = r2 + ddum(l)*ddum(l)
                        end do
                     end if
                     this%pairs_evaluated = this%pairs_evaluated + 1_8
                     if (ngbr(iip, jjp, r2, dum, ct) /= 0) then
                        id = nn(i, 1) + 1
                        call claim_column(id)
                        nn(i, 1) = id; nn(i, id) = j
                        nnmax = max(nnmax, id)
                        id = nn(j, 1) + 1
                        call claim_column(id)
                        nn(j, 1) = id; nn(j, id) = i
                        nnmax = max(nnmax, id)
                     end if
                     j = next(j)
                  end do
               end do
            end do
         end do
         icell = 1 + c(1) + nc(1)*(c(2) + nc(2)*c(3))
         next(i) = head(icell)
         head(icell) = i
      end do
      ! ascending neighbour order, as the reference's visiting order produces it
      do i = 1, nat
         do k = 3, nn(i, 1)
            t = nn(i, k)
            j = k - 1
            do while (j >= 2)
               if (nn(i, j) <= t) exit
               nn(i, j + 1) = nn(i, j)
               j = j - 1
            end do
            nn(i, j + 1) = t
         end do
      end do
      call claim_column(min(nm, nnmax + 1))           ! structb copies columns 1 .. nnmax + 1 (:1858-1860)
      if (nnmax == 0) call claim_column(min(nm, 2))
      nm = nnmax
      deallocate (cell_of, head, next)

   contains

      !> first use of column `col`: zero it for every atom (and every column before it); stop as the reference does on overflow
      subroutine claim_column(col)
         integer, intent(in) :: col
         if (col > nm) then
            write (6, '(" TOO MANY NEIGHBOURS")')
            write (6, '(" NEIGHBOUR MAP AS FAR AS", i6, "TH SITE")') i
            write (6, *) col, col, nm
            stop
         end if
         do while (ncol < col)
            ncol = ncol + 1
            nn(1:nat, ncol) = 0
         end do
      end subroutine claim_column

   end subroutine nncal_cells

   pure function cross(u, v) result(x)
      real(rp), intent(in) :: u(3), v(3)
      real(rp) :: x(3)
      x = [u(2)*v(3) - u(3)*v(2), u(3)*v(1) - u(1)*v(3), u(1)*v(2) - u(2)*v(1)]
   end function cross

   !> index of the cell `d` steps from cell `c` (0-based) in a direction with `n` cells; -1 outside a non-periodic direction
   pure integer function neighbour_cell(c, d, n, periodic) result(r)
      integer, intent(in) :: c, d, n
      logical, intent(in) :: periodic
      r = c + d
      if (periodic) then
         r = modulo(r, n)
      else if (r < 0 .or. r >= n) then
         r = -1
      end if
   end function neighbour_cell

end module lattice_cells_mod
