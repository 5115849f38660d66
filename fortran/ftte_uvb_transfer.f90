! ftte_uvb_transfer.f90 -- the drop-in: what replaces equiSources.f90:1383-1806 (everything after the
! computeOpacities loop inside `if (runUVBTransfer)`) in the reference driver.
!
! It is compiled TOGETHER WITH the reference (it uses the reference's `definitions` module for zoneType,
! baseGrid, uvb1..3, physicalBoxSize, nAngularLevel): see INTEGRATION.md for the two-line change to the
! reference's Makefile and main program.  This repository compiles it only as a syntax/interface check
! against oracle/_ref/definitions.mod when that has been built (fortran/Makefile: target `dropin-check`).
!
!   call ftteRunUVBTransfer(nx)        ! baseGrid holds kappa1..3 on entry, Jmean1..3 on return
!
! flatten tree -> C ABI -> scatter J: the cell-array order of writeCell (equiSources.f90:4044-4079).
module ftte_uvb_transfer

  use, intrinsic :: iso_c_binding
  use definitions
  use ftte_binding
  implicit none

  type(c_ptr), save, private :: ctx = c_null_ptr
  ! Several GPUs under the one serial host: allocate and fill before the first call, e.g. ftteDevices = (/ (i, i = 0, 7) /);
  ! the library then splits the frequency groups and the directions over them and sums J itself (include/ftte.h: ftte_create).
  integer(c_int), allocatable, target, save, public :: ftteDevices(:)
  integer(c_int64_t), private :: cursor
  ! The flattened cell array is kept from call to call (the reference's tree is static over a run) and pinned, so that
  ! kappa and J cross PCIe by DMA straight from / into these arrays; the library for its part keeps the tree, the sweep
  ! plan and the segment forests when ftte_set_grid sees the level list it already holds.
  integer(c_int32_t), allocatable, target, save, private :: lev(:)
  real(c_double), allocatable, target, save, private :: kap(:,:), Jflat(:,:)

contains

  subroutine ftteRunUVBTransfer(nx)
    integer, intent(in) :: nx
    integer(c_int64_t) :: ncell
    real(c_double), allocatable :: phi(:), theta(:), w(:)
    real(c_double) :: uvb(3)
    integer :: i, j, k, ndir, nside
    integer(kind=8) :: iray

    if (.not. c_associated(ctx)) then
       if (allocated(ftteDevices)) then
          call ftteCheck(c_null_ptr, ftte_create(ctx, size(ftteDevices), c_loc(ftteDevices)), 'ftte_create')
       else
          call ftteCheck(c_null_ptr, ftte_create(ctx, 1, c_null_ptr), 'ftte_create')
       endif
    endif

    ncell = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call countLeaves(baseGrid%cell(i,j,k), ncell)
          enddo
       enddo
    enddo
    if (allocated(lev)) then
       if (size(lev, kind=c_int64_t) /= ncell) then
          call ftteCheck(ctx, ftte_host_unregister(ctx, c_loc(kap)), 'ftte_host_unregister')
          call ftteCheck(ctx, ftte_host_unregister(ctx, c_loc(Jflat)), 'ftte_host_unregister')
          deallocate(lev, kap, Jflat)
       endif
    endif
    if (.not. allocated(lev)) then
       allocate(lev(ncell), kap(ncell,3), Jflat(ncell,3))
       call ftteCheck(ctx, ftte_host_register(ctx, c_loc(kap), int(24*ncell, c_size_t)), 'ftte_host_register')
       call ftteCheck(ctx, ftte_host_register(ctx, c_loc(Jflat), int(24*ncell, c_size_t)), 'ftte_host_register')
    endif
    cursor = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call gather(baseGrid%cell(i,j,k), 0, lev, kap)
          enddo
       enddo
    enddo

    ! direction list of the reference loop, equiSources.f90:1385-1391
    nside = 2**(nAngularLevel-1)
    ndir = 12*4**(nAngularLevel-1)
    allocate(phi(ndir), theta(ndir), w(ndir))
    do iray = 0, ndir-1
       if (ftte_pix2ang_nest(nside, iray, phi(iray+1), theta(iray+1)) /= FTTE_OK) stop 'ipix out of range'
    enddo
    w = 1./float(ndir)   ! the reference's single-precision quotient, :1386
    uvb = (/ uvb1, uvb2, uvb3 /)

    call ftteCheck(ctx, ftte_set_grid(ctx, nx, nx, nx, ncell, lev, physicalBoxSize), 'ftte_set_grid')
    call ftteCheck(ctx, ftte_diffuse_iteration(ctx, 3, kap, ndir, phi, theta, w, uvb, Jflat), 'ftte_diffuse_iteration')

    cursor = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call scatter(baseGrid%cell(i,j,k), Jflat)
          enddo
       enddo
    enddo
  end subroutine ftteRunUVBTransfer

  recursive subroutine countLeaves(c, total)
    type(zoneType) :: c
    integer(c_int64_t), intent(inout) :: total
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call countLeaves(c%cell(a,b,d), total)
             enddo
          enddo
       enddo
    else
       total = total + 1
    endif
  end subroutine countLeaves

  recursive subroutine gather(c, level, lev, kap)
    type(zoneType) :: c
    integer, intent(in) :: level
    integer(c_int32_t), intent(inout) :: lev(:)
    real(c_double), intent(inout) :: kap(:,:)
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call gather(c%cell(a,b,d), level+1, lev, kap)
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       lev(cursor) = level
       kap(cursor,1) = c%kappa1
       kap(cursor,2) = c%kappa2
       kap(cursor,3) = c%kappa3
    endif
  end subroutine gather

  recursive subroutine scatter(c, J)
    type(zoneType) :: c
    real(c_double), intent(in) :: J(:,:)
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call scatter(c%cell(a,b,d), J)
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       c%Jmean1 = J(cursor,1)
       c%Jmean2 = J(cursor,2)
       c%Jmean3 = J(cursor,3)
    endif
  end subroutine scatter

end module ftte_uvb_transfer
