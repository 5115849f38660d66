! ftte_stellar_transfer.f90 -- the drop-in for the point-source block: what replaces the star loop
! equiSources.f90:1260-1362 inside `if (runStellarTransfer)` in the reference driver, its per-star `src:` line (escape fractions
! at the seven output radii, :1342-1357) and the accumulation of cosmicSpectrum (:1350-1351) included.
!
! Compiled TOGETHER WITH the reference (modules `definitions` and `dust`); this repository compiles it only as an
! interface check against oracle/_ref/*.mod (fortran/Makefile: target `dropin-check`).
!
!   call setZeroRates(...)                                   ! unchanged, the reference zeroes the cells itself
!   call ftteRunStellarTransfer(nx, nStars, star, iSpectrum, coefSpectrum)
!
! On return krate24..26 and crate24..26 of every leaf hold what the reference's startNewLongRay deposits, highestPixelLevel and
! (per star, as after the reference's loop the last star's) ndotRemaining, ndotBoundary, ndotDust, ndotSpectrum, fraction of
! module localDefinitions / definitions are set, and cosmicSpectrum has received every star's share; the caller zeroes
! cosmicSpectrum before (:1258) and divides by nStarsSpecificAge after (:1366), as the reference does around its loop.
! localDefinitions is a module of the reference's main file: compile this file after it (INTEGRATION.md).
module ftte_stellar_transfer

  use, intrinsic :: iso_c_binding
  use definitions
  use localDefinitions
  use dust
  use ftte_binding
  implicit none

  type(c_ptr), save, private :: ctx = c_null_ptr
  integer(c_int64_t), private :: cursor

contains

  subroutine ftteRunStellarTransfer(nx, nStars, star, iSpectrum, coefSpectrum)
    integer, intent(in) :: nx, nStars, iSpectrum
    type(starType), intent(in) :: star(:)
    real(kind=RealKind), intent(in) :: coefSpectrum
    integer(c_int64_t) :: ncell, host(1)
    integer(c_int32_t), allocatable :: lev(:), pos(:)
    real(c_double), allocatable :: med(:,:), rates(:,:)
    real(c_double) :: ndot(1), totalIntegral, tmp, coefMetal
    real(c_double) :: escRemaining(nradius), escBoundary(nradius), escDust(1), escSpectrum(nenergy), escFraction(nradius)
    integer(c_int) :: highest
    integer :: i, j, k, iStar, iMetal

    if (.not. c_associated(ctx)) call ftteCheck(c_null_ptr, ftte_create(ctx, 1, c_null_ptr), 'ftte_create')

    ncell = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call countCells(baseGrid%cell(i,j,k), ncell)
          enddo
       enddo
    enddo
    allocate(lev(ncell), med(ncell,5), rates(ncell,6))
    cursor = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call gatherMedium(baseGrid%cell(i,j,k), 0, lev, med)
          enddo
       enddo
    enddo

    call ftteCheck(ctx, ftte_set_grid(ctx, nx, nx, nx, ncell, lev, physicalBoxSize), 'ftte_set_grid')
    call ftteCheck(ctx, ftte_set_medium(ctx, med(:,1), med(:,2), med(:,3), med(:,4), med(:,5), dustApproximation), &
         'ftte_set_medium')
    call ftteCheck(ctx, ftte_set_zero_rates(ctx), 'ftte_set_zero_rates')

    do iStar = 1, nStars
       if (star(iStar)%weight .gt. 0) then
          allocate(pos(3*star(iStar)%level+3))
          pos = star(iStar)%position(1:3*star(iStar)%level+3)
          call ftteCheck(ctx, ftte_locate_cell(ctx, star(iStar)%level, pos, host(1)), 'ftte_locate_cell')
          deallocate(pos)

          ! the population of this star: metallicity of its host cell, equiSources.f90:1281-1291
          if (med(host(1)+1,5) .gt. 1.e-20) then
             tmp = dlog10(med(host(1)+1,5))
          else
             tmp = - 20.
          endif
          iMetal = 1
          do while (tmp .gt. metallicity(iMetal+1))
             iMetal = iMetal + 1
             if (iMetal+1 .eq. nMetallicity) exit
          enddo
          coefMetal = (tmp-metallicity(iMetal))/(metallicity(iMetal+1)-metallicity(iMetal))
          coefMetal = min(max(0.d0,coefMetal),1.d0)

          call ftteCheck(ctx, ftte_stellar_beta_table(ctx, a_smc, nWavelengths, wavelength, nSpectra, nMetallicity, &
               specificLuminosity, iSpectrum, coefSpectrum, iMetal, coefMetal, totalIntegral), 'ftte_stellar_beta_table')

          ndot(1) = float(star(iStar)%weight)       ! :1303
          call ftteCheck(ctx, ftte_point_sources(ctx, 1, host, ndot, highest), 'ftte_point_sources')
          highestPixelLevel = highest

          ! what the tracer kept for this star (:3198-3233, :3336-3345), the src: line (:1342-1357) and the cosmic spectrum
          call ftteCheck(ctx, ftte_point_escape(ctx, 1, escRemaining, escBoundary, escDust, escSpectrum, escFraction), 'ftte_point_escape')
          ndotRemaining = escRemaining
          ndotBoundary = escBoundary
          ndotDust = escDust(1)
          ndotSpectrum = escSpectrum
          fraction = escFraction
          cosmicSpectrum = cosmicSpectrum + float(star(iStar)%weight) * ndotSpectrum/(ndot(1)-ndotBoundary(nradius))
          write(*,1015) iStar, star(iStar)%level, med(host(1)+1,1) * mh / (psi * med(host(1)+1,4)), &
               highestPixelLevel, fraction, star(iStar)%weight
1015      format('src: ', i5, i3, es13.5, i3, 7f9.5, i8)
       endif
    enddo

    call ftteCheck(ctx, ftte_get_point_rates(ctx, rates), 'ftte_get_point_rates')
    cursor = 0
    do i = 1, nx
       do j = 1, nx
          do k = 1, nx
             call scatterRates(baseGrid%cell(i,j,k), rates)
          enddo
       enddo
    enddo
  end subroutine ftteRunStellarTransfer

  recursive subroutine countCells(c, total)
    type(zoneType) :: c
    integer(c_int64_t), intent(inout) :: total
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call countCells(c%cell(a,b,d), total)
             enddo
          enddo
       enddo
    else
       total = total + 1
    endif
  end subroutine countCells

  recursive subroutine gatherMedium(c, level, lev, med)
    type(zoneType) :: c
    integer, intent(in) :: level
    integer(c_int32_t), intent(inout) :: lev(:)
    real(c_double), intent(inout) :: med(:,:)
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call gatherMedium(c%cell(a,b,d), level+1, lev, med)
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       lev(cursor) = level
       med(cursor,1) = c%HI
       med(cursor,2) = c%HeI
       med(cursor,3) = c%HeII
       med(cursor,4) = c%rho
       med(cursor,5) = c%abun2
    endif
  end subroutine gatherMedium

  recursive subroutine scatterRates(c, rates)
    type(zoneType) :: c
    real(c_double), intent(in) :: rates(:,:)
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call scatterRates(c%cell(a,b,d), rates)
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       c%krate24 = c%krate24 + rates(cursor,1)
       c%krate25 = c%krate25 + rates(cursor,2)
       c%krate26 = c%krate26 + rates(cursor,3)
       c%crate24 = c%crate24 + rates(cursor,4)
       c%crate25 = c%crate25 + rates(cursor,5)
       c%crate26 = c%crate26 + rates(cursor,6)
    endif
  end subroutine scatterRates

end module ftte_stellar_transfer
