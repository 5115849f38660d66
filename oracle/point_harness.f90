! point_harness.f90 -- TEST INFRASTRUCTURE ONLY (never linked into the product).
!
! Drives the reference's own point-source code on a synthetic cell array:
!   * stellarBetaTable (stellarBetaTable.f90), dustCrossSection (dustModule.f90), stellarPopulation
!     (stellarPopulationModule.f90): compiled where they lie under /root/reference;
!   * startNewLongRay, drawSegment, find/zoom??Neighbour, absoluteCoordinates, localizeSplitContinuationCell,
!     getRatesHydrogenHelium, setZeroRates, pix2ang_nest (+ rotateAngles, getAngle, arcsin, mk_pix2xy):
!     contained procedures of the reference's main program (equiSources.f90), which oracle/Makefile lifts at build
!     time -- by line range, through a temporary file outside the repository -- into the module `pointExtract`
!     (object and .mod in oracle/_ref/ only).
! What the main program does around them (equiSources.f90:296-309 rmax, :1256-1329 the per-source loop) is restated
! here; of the escape bookkeeping the tracer accumulates (ndotRemaining, ndotBoundary, ndotDust, ndotSpectrum, :3198-3233,
! :3336-3345) per star, and the `fraction` the main program prints in its `src:` line (:1342-1348), only that quotient is
! written out here.
!
! usage: point_harness <case.bin> <out.bin>
! case.bin (stream):
!   int32 n, ncell, nsrc, dust, nsample, npixlevel ; real64 box
!   int32 level(ncell) ; real64 HI, HeI, HeII, rho, abun2 (ncell each)
!   int32 srcLeaf(nsrc) (1-based cell-array index), srcWeight(nsrc)
!   real64 a_smc(7,5) ; real64 wavelength(1221) [cm] ; real64 specificLuminosity(5,37,1221)
!   int32 iSpectrum, iMetal ; real64 coefSpectrum, coefMetal
!   real64 sample(4,nsample)  (tau1,tau2,tau3,tauDust for getRatesHydrogenHelium)
! out.bin:
!   real64 totalIntegral ; real64 reactionRate1..3, energyRate1..3 (11^4 each, Fortran order)
!   real64 outputSigma24, 25, 26, Dust (300 each)
!   real64 rates(2,3,nsample)    (numberRate, heatingRate) x reaction
!   real64 pix(2, 12*4^(L-1)) for L = 1..npixlevel    (phi, theta of pix2ang_nest)
!   real64 rmax(30)
!   real64 krate24, krate25, krate26, crate24, crate25, crate26 (ncell each) ; int32 highestPixelLevel
!   real64 ndotRemaining(7,nsrc), ndotBoundary(7,nsrc), ndotDust(nsrc), ndotSpectrum(300,nsrc), fraction(7,nsrc)
program point_harness

  use definitions
  use localDefinitions
  use dust
  use pointExtract

  implicit none

  integer :: n, ncell, nsrc, nsample, npixlevel, ios, cursor, is, ir, L, iSpectrum, iMetal
  integer*8 :: ipix, iray8
  integer, allocatable :: lev(:), srcLeaf(:), srcWeight(:)
  real(kind=RealKind), allocatable :: fHI(:), fHeI(:), fHeII(:), frho(:), fabun(:), sample(:,:), rates(:,:,:), &
       pix(:,:), kout(:,:), escRemaining(:,:), escBoundary(:,:), escDust(:), escSpectrum(:,:), escFraction(:,:)
  real(kind=RealKind) :: box, coefSpectrum, coefMetal, totalIntegral, ndot1
  character(len=512) :: caseName, outName
  integer :: bi, bj, bk, want, pathLen, iradius
  integer(kind=8) :: tick0, tick1, tickRate
  integer, target :: path(33)
  type(zoneType), pointer :: host
  type(pixelType), target :: sphere
  type(pixelType), pointer :: leafPixel
  type(pointType) :: startingPoint
  logical :: found

  call get_command_argument(1, caseName)
  call get_command_argument(2, outName)
  open(11, file=trim(caseName), access='stream', form='unformatted', status='old', iostat=ios)
  if (ios /= 0) stop 'point_harness: cannot open case file'
  read(11) n, ncell, nsrc, dustApproximation, nsample, npixlevel
  read(11) box
  allocate(lev(ncell), fHI(ncell), fHeI(ncell), fHeII(ncell), frho(ncell), fabun(ncell))
  allocate(srcLeaf(nsrc), srcWeight(nsrc), sample(4,nsample), rates(2,3,nsample))
  read(11) lev
  read(11) fHI, fHeI, fHeII, frho, fabun
  read(11) srcLeaf, srcWeight
  read(11) a_smc
  read(11) wavelength
  read(11) specificLuminosity
  read(11) iSpectrum, iMetal
  read(11) coefSpectrum, coefMetal
  read(11) sample
  close(11)

  physicalBoxSize = box

  ! equiSources.f90:296-309 (the literal table is overwritten by the formula before use)
  do ir = 1, nrmax
     rmax(ir) = sqrt(3.)*(sqrt(0.5*4.**(ir-1)-1./12.)+0.5)
  enddo
  rmax = rmax/2.

  ! ---- tree from the leaf list (readCellArray.f90:154-187), storage indexing as the point-source tracer uses it
  baseGrid%refined = .true.
  baseGrid%level = -1
  allocate(baseGrid%cell(n,n,n))
  cursor = 0
  do bi = 1, n
     do bj = 1, n
        do bk = 1, n
           baseGrid%cell(bi,bj,bk)%parent => baseGrid
           call growCell(baseGrid%cell(bi,bj,bk), 0)
        enddo
     enddo
  enddo
  if (cursor /= ncell) stop 'point_harness: level list does not describe a tree of ncell leaves'

  open(12, file=trim(outName), access='stream', form='unformatted', status='replace')

  ! ---- P3: the rate tables of this population
  call stellarBetaTable(nfbins, frequencyBinWidth, totalIntegral, iSpectrum, coefSpectrum, iMetal, coefMetal)
  write(12) totalIntegral
  write(12) reactionRate1, reactionRate2, reactionRate3, energyRate1, energyRate2, energyRate3
  write(12) outputSigma24, outputSigma25, outputSigma26, outputSigmaDust

  ! ---- P2: table look-ups
  do is = 1, nsample
     do ir = 1, 3
        call getRatesHydrogenHelium(ir, sample(1,is), sample(2,is), sample(3,is), sample(4,is), rates(1,ir,is), rates(2,ir,is))
     enddo
  enddo
  write(12) rates

  ! ---- A2: pixel centres
  do L = 1, npixlevel
     allocate(pix(2, 12*4**(L-1)))
     do ipix = 0, 12*4**(L-1) - 1
        call pix2ang_nest(2**(L-1), ipix, pix(1,ipix+1), pix(2,ipix+1))
     enddo
     write(12) pix
     deallocate(pix)
  enddo
  write(12) rmax

  ! ---- P1: the per-source loop, equiSources.f90:1260-1329
  do bi = 1, n
     do bj = 1, n
        do bk = 1, n
           call setZeroRates(baseGrid%cell(bi,bj,bk))
        enddo
     enddo
  enddo
  sphere%refined = .false.
  sphere%level = 0
  highestPixelLevel = 0
  allocate(escRemaining(nradius,nsrc), escBoundary(nradius,nsrc), escDust(nsrc), escSpectrum(nenergy,nsrc), escFraction(nradius,nsrc))
  call system_clock(tick0, tickRate)
  do is = 1, nsrc
     ndotRemaining = 0.
     ndotBoundary = 0.
     ndotDust = 0.
     ndotSpectrum = 0.
     startingPoint%x = 0.5
     startingPoint%y = 0.5
     startingPoint%z = 0.5
     ! the star's host leaf and its call sequence
     want = srcLeaf(is)
     cursor = 0
     found = .false.
     nullify(host)
     do bi = 1, n
        do bj = 1, n
           do bk = 1, n
              if (.not. found) then
                 path(1:3) = (/ bi, bj, bk /)
                 call findLeaf(baseGrid%cell(bi,bj,bk), 0)
              endif
           enddo
        enddo
     enddo
     if (.not. found) stop 'point_harness: source leaf not found'

     if (.not. sphere%refined) then
        allocate(sphere%pixel(12))
        sphere%refined = .true.
        sphere%pixel(:)%level = -99
     endif
     ndot1 = float(srcWeight(is))
     do iray8 = 1, 12
        leafPixel => sphere%pixel(iray8)
        if (leafPixel%level .ne. 1) then
           leafPixel%level = 1
           leafPixel%refined = .false.
           leafPixel%parent => sphere
           ipix = iray8 - 1
           call pix2ang_nest(1, ipix, leafPixel%phi, leafPixel%theta)
        endif
        call startNewLongRay(host, startingPoint, leafPixel, iray8, int(host%level), path(1:pathLen), 0.d0, &
             ndot1/12.d0, 0.d0, 0.d0, 0.d0, 0.d0, n, n, n)
     enddo
     ! what the tracer accumulated for this star, and the quotient of equiSources.f90:1342-1348
     escRemaining(:,is) = ndotRemaining
     escBoundary(:,is) = ndotBoundary
     escDust(is) = ndotDust
     escSpectrum(:,is) = ndotSpectrum
     do iradius = 1, nradius
        if (ndotBoundary(iradius).lt.1.) then
           fraction(iradius) = ndotRemaining(iradius)/(ndot1-ndotBoundary(iradius))
        else
           fraction(iradius) = 0.
        endif
     enddo
     escFraction(:,is) = fraction
  enddo

  call system_clock(tick1)
  write(*,'(a,f12.4)') ' TRACE_SECONDS ', dble(tick1-tick0)/dble(tickRate)

  allocate(kout(ncell,6))
  cursor = 0
  do bi = 1, n
     do bj = 1, n
        do bk = 1, n
           call harvest(baseGrid%cell(bi,bj,bk))
        enddo
     enddo
  enddo
  write(12) kout
  write(12) highestPixelLevel
  write(12) escRemaining, escBoundary, escDust, escSpectrum, escFraction
  close(12)

contains

  recursive subroutine growCell(c, level)
    type(zoneType), target :: c
    integer, intent(in) :: level
    integer :: a, b, d
    cursor = cursor + 1
    if (cursor > ncell) stop 'point_harness: ran past the end of the level list'
    nullify(c%cell)
    c%level = int(level,1)
    if (lev(cursor) == level) then
       c%refined = .false.
       c%HI = fHI(cursor)
       c%HeI = fHeI(cursor)
       c%HeII = fHeII(cursor)
       c%rho = frho(cursor)
       c%abun2 = fabun(cursor)
    else if (lev(cursor) > level) then
       cursor = cursor - 1
       c%refined = .true.
       allocate(c%cell(2,2,2))
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                c%cell(a,b,d)%parent => c
                call growCell(c%cell(a,b,d), level+1)
             enddo
          enddo
       enddo
    else
       stop 'point_harness: level list is not depth-first'
    endif
  end subroutine growCell

  recursive subroutine findLeaf(c, level)
    type(zoneType), target :: c
    integer, intent(in) :: level
    integer :: a, b, d
    if (found) return
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                if (.not. found) then
                   path(3*level+4:3*level+6) = (/ a, b, d /)
                   call findLeaf(c%cell(a,b,d), level+1)
                endif
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       if (cursor == want) then
          found = .true.
          host => c
          pathLen = 3*level + 3
       endif
    endif
  end subroutine findLeaf

  recursive subroutine harvest(c)
    type(zoneType) :: c
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call harvest(c%cell(a,b,d))
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       kout(cursor,1) = c%krate24
       kout(cursor,2) = c%krate25
       kout(cursor,3) = c%krate26
       kout(cursor,4) = c%crate24
       kout(cursor,5) = c%crate25
       kout(cursor,6) = c%crate26
    endif
  end subroutine harvest

end program point_harness
