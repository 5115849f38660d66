! ref_harness.f90 -- TEST INFRASTRUCTURE ONLY (never linked into the product).
!
! A driver written for this repository that links against the *reference's own*
! modules (definitions, rotateIndicesModule, transportRoutinesModule, compiled
! where they lie under /root/reference by oracle/Makefile into oracle/_ref/)
! and replays one diffuse-transfer iteration of the reference on a synthetic
! cell array, so that golden vectors come out of the reference's arithmetic
! rather than out of a restatement of it.
!
! What the reference does around these calls lives in the main program
! (equiSources.f90:1385-1806) and is not callable.  Two routes:
!   * dumpGeometry + 16 (what the goldens are made with): module driverExtract, the reference's own lines
!     :1393-1801 lifted by line range at build time (oracle/Makefile), INCLUDING the inline branch for unrefined
!     base cells (:1580-1788) -- nothing of the per-direction work is restated;
!   * otherwise this file's sweepOneDirection: the same sequence of steps (fold direction, per-layer pattern advance,
!     pattern attachment, neighbour linking, transport) written out here through the reference's public routines -- kept
!     as a cross-check (tests assert both routes give the same bits) and for the grid-less geometry dump;
!     it calls the reference's public
! routines: setPattern, setRaysRefined, localizeCellFindNeighbours, transport,
! patternNullify, rotateIndices.  Every leaf cell, refined or not, is pushed
! through the reference `transport` (transportRoutinesModule.f90:560), whose leaf
! branch carries the same arithmetic as the inlined base-cell code of
! equiSources.f90:1580-1788.
!
! usage:  ref_harness <case.bin> <out.bin>
!         ref_harness --rotate-table <out.bin>
!
! case.bin (stream, little endian):
!   int32  n, ncell, ndir, dumpGeometry
!   real64 box
!   real64 uvb(3)
!   int32  level(ncell)            cell-array (depth-first leaf) order
!   real64 kappa(ncell,3)          group-major
!   real64 phi(ndir), theta(ndir), w(ndir)   un-folded angles (phi in (0,2pi), theta in (-pi/2,pi/2))
! out.bin:
!   real64 J(ncell,3)
!   (dumpGeometry == 2: no J block, no sweep: geometry records only)
!   if dumpGeometry /= 0, per direction:
!       int32 izone; real64 phiFold, thetaFold
!       per base layer i=1..n: real64 xy(x0,y0,len), xz(x0,z0,len), yz(y0,z0,len)
!                              int32 xzActive, yzActive, xyTop, xzTop, yzTop
program ref_harness

  use definitions
  use rotateIndicesModule
  use transportRoutinesModule
  use driverExtract

  implicit none

  integer :: n, ncell, ndir, dumpGeometry, idir, cursor, ios
  logical :: liftedDriver
  integer, allocatable :: lev(:)
  real(kind=RealKind), allocatable :: kap(:,:), jout(:,:), phiIn(:), thetaIn(:), wIn(:)
  real(kind=RealKind) :: box, uvbIn(3)
  character(len=512) :: caseName, outName
  integer :: bi, bj, bk
  integer(kind=8) :: tick0, tick1, tickRate

  call get_command_argument(1, caseName)
  call get_command_argument(2, outName)

  if (trim(caseName) == '--rotate-table') then
     call dumpRotateTable()
     stop
  endif

  open(11, file=trim(caseName), access='stream', form='unformatted', status='old', iostat=ios)
  if (ios /= 0) stop 'ref_harness: cannot open case file'
  read(11) n, ncell, ndir, dumpGeometry
  ! dumpGeometry + 16: every direction goes through the reference's OWN driver lines (module driverExtract, lifted from
  ! equiSources.f90:1393-1801 by oracle/Makefile) instead of this file's sweepOneDirection
  liftedDriver = iand(dumpGeometry, 16) /= 0
  dumpGeometry = iand(dumpGeometry, 15)
  if (liftedDriver .and. dumpGeometry == 2) stop 'ref_harness: the lifted driver needs a grid (use dumpGeometry = 1)'
  read(11) box
  read(11) uvbIn
  allocate(lev(ncell), kap(ncell,3), jout(ncell,3), phiIn(ndir), thetaIn(ndir), wIn(ndir))
  read(11) lev
  read(11) kap
  read(11) phiIn, thetaIn, wIn
  close(11)

  physicalBoxSize = box
  uvb1 = uvbIn(1)
  uvb2 = uvbIn(2)
  uvb3 = uvbIn(3)

  ! dumpGeometry == 2: geometry records only (no grid, no sweep); ncell may be 0
  if (dumpGeometry == 2) then
     open(12, file=trim(outName), access='stream', form='unformatted', status='replace')
     do idir = 1, ndir
        call sweepOneDirection(phiIn(idir), thetaIn(idir), wIn(idir))
     enddo
     close(12)
     stop
  endif

  ! ---- tree from the flat leaf list (layout: readCellArray.f90:154-187) ----
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
  if (cursor /= ncell) stop 'ref_harness: level list does not describe a tree of ncell leaves'

  open(12, file=trim(outName), access='stream', form='unformatted', status='replace')

  call system_clock(tick0, tickRate)
  if (liftedDriver) then
     if (dumpGeometry /= 0) driverDumpUnit = 12
     do idir = 1, ndir
        call referenceDirection(phiIn(idir), thetaIn(idir), wIn(idir), n, n, n)
     enddo
  else
     do idir = 1, ndir
        call sweepOneDirection(phiIn(idir), thetaIn(idir), wIn(idir))
     enddo
  endif
  call system_clock(tick1)
  ! wall time of the direction loop alone (patterns + neighbour links + transport), for bench.py
  write(*,'(a,1x,es16.8)') 'SWEEP_SECONDS', dble(tick1-tick0)/dble(tickRate)

  cursor = 0
  do bi = 1, n
     do bj = 1, n
        do bk = 1, n
           call harvestCell(baseGrid%cell(bi,bj,bk))
        enddo
     enddo
  enddo

  ! J goes first in the file: rewrite from the start
  if (dumpGeometry /= 0) then
     close(12)
     call prependJ()
  else
     write(12) jout
     close(12)
  endif

contains

  recursive subroutine growCell(c, level)
    type(zoneType), target :: c
    integer, intent(in) :: level
    integer :: a, b, d
    cursor = cursor + 1
    if (cursor > ncell) stop 'ref_harness: ran past the end of the level list'
    nullify(c%cell)
    c%level = int(level,1)
    c%Jmean1 = 0.d0
    c%Jmean2 = 0.d0
    c%Jmean3 = 0.d0
    if (lev(cursor) == level) then
       c%refined = .false.
       c%kappa1 = kap(cursor,1)
       c%kappa2 = kap(cursor,2)
       c%kappa3 = kap(cursor,3)
       c%HI = 0.d0
       c%HeI = 0.d0
       c%HeII = 0.d0
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
       stop 'ref_harness: level list is not depth-first'
    endif
  end subroutine growCell

  recursive subroutine harvestCell(c)
    type(zoneType) :: c
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call harvestCell(c%cell(a,b,d))
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
       jout(cursor,1) = c%Jmean1
       jout(cursor,2) = c%Jmean2
       jout(cursor,3) = c%Jmean3
    endif
  end subroutine harvestCell

  subroutine foldDirection(phiLarge, thetaLarge, ang)
    ! same decisions as equiSources.f90:1395-1454 (quadrant of phi, sign of theta,
    ! dominant axis), same constants (pi of the definitions module)
    real(kind=RealKind), intent(in) :: phiLarge, thetaLarge
    type(angleType), intent(out) :: ang
    real(kind=RealKind) :: p1, t1, invz, invx, invy
    integer :: zone, q

    zone = 1
    q = -1
    if (phiLarge > 0.     .and. phiLarge < 0.5*pi) q = 0
    if (phiLarge > 0.5*pi .and. phiLarge < pi)     q = 1
    if (phiLarge > pi     .and. phiLarge < 1.5*pi) q = 2
    if (phiLarge > 1.5*pi .and. phiLarge < 2.*pi)  q = 3
    if (q < 0) stop 'ref_harness: phi on a quadrant boundary'
    p1 = phiLarge - q*0.5*pi
    if (q == 2) p1 = phiLarge - pi
    zone = zone + 3*q

    if (thetaLarge > 0. .and. thetaLarge < 0.5*pi) then
       t1 = thetaLarge
    else if (thetaLarge > -0.5*pi .and. thetaLarge < 0.) then
       t1 = -thetaLarge
       zone = zone + 12
    else
       stop 'ref_harness: theta on a boundary'
    endif

    invz = 1./sin(t1)
    invx = 1./(cos(p1)*cos(t1))
    invy = 1./(sin(p1)*cos(t1))

    if (invz < min(invx,invy)) then
       ang%theta = t1
       ang%phi = p1
    else if (invx < min(invz,invy)) then
       ang%theta = clampedAsin(cos(t1)*cos(p1))
       ang%phi = clampedAsin(sin(t1)/cos(ang%theta))
       zone = zone + 1
    else if (invy < min(invz,invx)) then
       ang%theta = clampedAsin(cos(t1)*sin(p1))
       ang%phi = acos(sin(t1)/cos(ang%theta))
       zone = zone + 2
    else
       stop 'ref_harness: tie between dominant axes'
    endif
    ang%izone = int(zone,1)
  end subroutine foldDirection

  function clampedAsin(x) result(a)
    ! equiSources.f90:2277-2295
    real(kind=RealKind), intent(in) :: x
    real(kind=RealKind) :: a
    if (x > 1.d0) then
       a = halfPi
    else if (x < -1.d0) then
       a = -halfPi
    else
       a = asin(x)
    endif
  end function clampedAsin

  subroutine sweepOneDirection(phiLarge, thetaLarge, weight)
    real(kind=RealKind), intent(in) :: phiLarge, thetaLarge, weight
    type(angleType) :: ang
    type(patternType), allocatable, target :: layer(:)
    integer, dimension(2,2,2) :: is, js, ks
    integer :: i, j, k, ic, jc, kc
    real(kind=RealKind) :: cphi, sphi, cth, tth, delta
    type(zoneType), pointer :: c

    call foldDirection(phiLarge, thetaLarge, ang)

    do i = 1, 2
       do j = 1, 2
          do k = 1, 2
             call rotateIndices(i,j,k,2,2,2,ang%izone,is(i,j,k),js(i,j,k),ks(i,j,k))
          enddo
       enddo
    enddo

    allocate(layer(n))

    ! per-layer geometry: the entry point of layer i is where the top-ending
    ! segment of layer i-1 leaves the unit cell (equiSources.f90:1495-1534)
    do i = 1, n
       layer(i)%refined = .false.
       nullify(layer(i)%cell)
       if (i == 1) then
          layer(i)%xyRay%x0 = 0.5
          layer(i)%xyRay%y0 = 0.5
       else
          select case (layer(i-1)%xyTop)
          case (xyEnd)
             layer(i)%xyRay%x0 = layer(i-1)%xyRay%x0 + cos(ang%phi)/tan(ang%theta)
             layer(i)%xyRay%y0 = layer(i-1)%xyRay%y0 + sin(ang%phi)/tan(ang%theta)
          case (xzEnd)
             layer(i)%xyRay%x0 = layer(i-1)%xzRay%x0 + layer(i-1)%xzRay%len*cos(ang%theta)*cos(ang%phi)
             layer(i)%xyRay%y0 = layer(i-1)%xzRay%len*cos(ang%theta)*sin(ang%phi)
          case (yzEnd)
             layer(i)%xyRay%x0 = layer(i-1)%yzRay%len*cos(ang%theta)*cos(ang%phi)
             layer(i)%xyRay%y0 = layer(i-1)%yzRay%y0 + layer(i-1)%yzRay%len*cos(ang%theta)*sin(ang%phi)
          case default
             stop 'ref_harness: layer without a top-ending segment'
          end select
          if (layer(i)%xyRay%x0 > 1. .or. layer(i)%xyRay%y0 > 1.) stop 'ref_harness: entry point left the unit cell'
       endif
       call setPattern(layer(i), ang%phi, ang%theta)
    enddo

    if (dumpGeometry /= 0) then
       write(12) int(ang%izone,4), ang%phi, ang%theta
       do i = 1, n
          write(12) layer(i)%xyRay%x0, layer(i)%xyRay%y0, layer(i)%xyRay%len
          write(12) layer(i)%xzRay%x0, layer(i)%xzRay%z0, layer(i)%xzRay%len
          write(12) layer(i)%yzRay%y0, layer(i)%yzRay%z0, layer(i)%yzRay%len
          write(12) merge(1,0,logical(layer(i)%xzRayActive)), merge(1,0,logical(layer(i)%yzRayActive)), &
               int(layer(i)%xyTop,4), int(layer(i)%xzTop,4), int(layer(i)%yzTop,4)
       enddo
    endif

    if (dumpGeometry == 2) then
       deallocate(layer)
       return
    endif

    ! attach patterns (and build the per-layer pattern trees under refined cells)
    do i = 1, n
       do j = 1, n
          do k = 1, n
             call rotateIndices(i,j,k,n,n,n,ang%izone,ic,jc,kc)
             c => baseGrid%cell(ic,jc,kc)
             c%pattern => layer(i)
             if (c%refined) call setRaysRefined(c, layer(i), is, js, ks, ang%phi, ang%theta)
             c%parent => baseGrid
          enddo
       enddo
    enddo

    ! upstream-cell links for this direction
    do i = 1, n
       do j = 1, n
          do k = 1, n
             call rotateIndices(i,j,k,n,n,n,ang%izone,ic,jc,kc)
             call localizeCellFindNeighbours(baseGrid%cell(ic,jc,kc), 0, (/i,j,k/), is, js, ks, n, n, n, ang%izone)
          enddo
       enddo
    enddo

    ! the sweep itself, reference order: march axis outermost
    delta = physicalBoxSize/dfloat(n)
    do i = 1, n
       do j = 1, n
          do k = 1, n
             call rotateIndices(i,j,k,n,n,n,ang%izone,ic,jc,kc)
             call transport(baseGrid%cell(ic,jc,kc), weight, is, js, ks, delta)
          enddo
       enddo
    enddo

    do i = 1, n
       if (layer(i)%refined) call patternNullify(layer(i))
    enddo
    deallocate(layer)
  end subroutine sweepOneDirection

  subroutine dumpRotateTable()
    ! rotateIndices (rotateIndicesModule.f90:7) on a 3 x 4 x 5 index box inside storage
    ! extents (7,11,13), all 24 zones: int32 (icell,jcell,kcell) per (izone,i,j,k), k fastest
    integer :: z, i, j, k, ic, jc, kc
    open(12, file=trim(outName), access='stream', form='unformatted', status='replace')
    do z = 1, 24
       do i = 1, 3
          do j = 1, 4
             do k = 1, 5
                call rotateIndices(i, j, k, 7, 11, 13, int(z,1), ic, jc, kc)
                write(12) ic, jc, kc
             enddo
          enddo
       enddo
    enddo
    close(12)
  end subroutine dumpRotateTable

  subroutine prependJ()
    ! geometry records were streamed while sweeping; J is known only at the end.
    ! Rewrite the file as  J | geometry.
    integer(kind=8) :: nbytes
    character(len=1), allocatable :: buf(:)
    inquire(file=trim(outName), size=nbytes)
    allocate(buf(nbytes))
    open(12, file=trim(outName), access='stream', form='unformatted', status='old')
    read(12) buf
    close(12)
    open(12, file=trim(outName), access='stream', form='unformatted', status='replace')
    write(12) jout
    write(12) buf
    close(12)
  end subroutine prependJ

end program ref_harness
