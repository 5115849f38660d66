! ftte_rate_equations.f90 -- the drop-in for the chemistry step: what replaces the solveRateEquations calls of the loop
! equiSources.f90:1824-1831 in the reference driver (computeMass stays where it is).
!
! Compiled TOGETHER WITH the reference (module `definitions`); this repository compiles it only as an interface check
! against oracle/_ref/definitions.mod (fortran/Makefile: target `dropin-check`).
!
!   call ftteSolveRateEquations(nx, runUVBTransfer)
!
! On entry the leaves hold rho, tgas, HI, HeI, HeII, krate24..26 (the point-source block) and Jmean1..3 (the diffuse
! block); on return HI, HeI, HeII are the reference's new equilibrium values.
module ftte_rate_equations

  use, intrinsic :: iso_c_binding
  use definitions
  use ftte_binding
  implicit none

  type(c_ptr), save, private :: ctx = c_null_ptr
  integer(c_int64_t), private :: cursor

contains

  subroutine ftteSolveRateEquations(nx, runUVBTransfer)
    integer, intent(in) :: nx
    logical, intent(in) :: runUVBTransfer
    integer(c_int64_t) :: ncell
    integer(c_int32_t), allocatable :: lev(:)
    real(c_double), allocatable :: f(:,:), rates(:,:)
    real(c_double) :: ksi(3,3), uniform(3), change
    integer(c_int) :: uvb
    integer :: i, j, k

    if (.not. c_associated(ctx)) call ftteCheck(c_null_ptr, ftte_create(ctx, 1, c_null_ptr), 'ftte_create')

    ncell = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call countChemCells(baseGrid%cell(i,j,k), ncell)
          enddo
       enddo
    enddo
    allocate(lev(ncell), f(ncell,11), rates(ncell,6))
    cursor = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call gatherState(baseGrid%cell(i,j,k), 0, lev, f)
          enddo
       enddo
    enddo
    rates = 0.d0
    rates(:,1:3) = f(:,6:8)

    ksi(:,1) = (/ group1%ksi24, group1%ksi25, group1%ksi26 /)
    ksi(:,2) = (/ group2%ksi24, group2%ksi25, group2%ksi26 /)
    ksi(:,3) = (/ group3%ksi24, group3%ksi25, group3%ksi26 /)
    uniform = (/ uniformQuasar*quasar%ksi24 + uniformStellar*stellar%ksi24, &
                 uniformQuasar*quasar%ksi25 + uniformStellar*stellar%ksi25, &
                 uniformQuasar*quasar%ksi26 + uniformStellar*stellar%ksi26 /)
    uvb = 0
    if (runUVBTransfer) uvb = 1

    call ftteCheck(ctx, ftte_set_grid(ctx, nx, nx, nx, ncell, lev, physicalBoxSize), 'ftte_set_grid')
    call ftteCheck(ctx, ftte_set_rate_coefficients(ctx, nratec, logtem0, logtem9, dlogtem, k1a, k2a, k3a, k4a, k5a, k6a), &
         'ftte_set_rate_coefficients')
    call ftteCheck(ctx, ftte_set_medium(ctx, f(:,3), f(:,4), f(:,5), f(:,1), f(:,1), 0), 'ftte_set_medium')
    call ftteCheck(ctx, ftte_set_temperature(ctx, f(:,2)), 'ftte_set_temperature')
    call ftteCheck(ctx, ftte_set_point_rates(ctx, rates), 'ftte_set_point_rates')
    call ftteCheck(ctx, ftte_solve_rate_equations(ctx, uvb, f(:,9:11), ksi, uniform, selfShieldingThreshold, 1, change), &
         'ftte_solve_rate_equations')
    call ftteCheck(ctx, ftte_get_medium(ctx, f(:,3), f(:,4), f(:,5)), 'ftte_get_medium')

    cursor = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call scatterState(baseGrid%cell(i,j,k), f)
          enddo
       enddo
    enddo
  end subroutine ftteSolveRateEquations

  recursive subroutine countChemCells(c, total)
    type(zoneType) :: c
    integer(c_int64_t), intent(inout) :: total
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call countChemCells(c%cell(a,b,d), total)
             enddo
          enddo
       enddo
    else
       total = total + 1
    endif
  end subroutine countChemCells

  recursive subroutine gatherState(c, level, lev, f)
    type(zoneType) :: c
    integer, intent(in) :: level
    integer(c_int32_t), intent(inout) :: lev(:)
    real(c_double), intent(inout) :: f(:,:)
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call gatherState(c%cell(a,b,d), level+1, lev, f)
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       lev(cursor) = level
       f(cursor,1) = c%rho
       f(cursor,2) = c%tgas
       f(cursor,3) = c%HI
       f(cursor,4) = c%HeI
       f(cursor,5) = c%HeII
       f(cursor,6) = c%krate24
       f(cursor,7) = c%krate25
       f(cursor,8) = c%krate26
       f(cursor,9) = c%Jmean1
       f(cursor,10) = c%Jmean2
       f(cursor,11) = c%Jmean3
    endif
  end subroutine gatherState

  recursive subroutine scatterState(c, f)
    type(zoneType) :: c
    real(c_double), intent(in) :: f(:,:)
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call scatterState(c%cell(a,b,d), f)
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       c%HI = f(cursor,3)
       c%HeI = f(cursor,4)
       c%HeII = f(cursor,5)
    endif
  end subroutine scatterState

end module ftte_rate_equations
