! dropin_check.f90 -- TEST INFRASTRUCTURE.  The drop-ins of fortran/ (ftte_uvb_transfer, ftte_rate_equations) exercised the
! way the reference driver would: on the reference's own data structures (module `definitions`, the fully threaded tree
! under baseGrid), linked against libftte.so.  Needs the reference's definitions module as built in oracle/_ref.
!
!   dropin_check uvb  <case.bin> <out.bin>     case: int32 n, ncell ; real64 box, uvb(3) ; int32 level(ncell) ; real64 kappa(ncell,3)
!                                              out : real64 Jmean(ncell,3) after ftteRunUVBTransfer(n)
!   dropin_check chem <case.bin> <out.bin>     case: int32 n, ncell, runUVB ; real64 box ; int32 level ; real64 f(ncell,11) (rho, tgas,
!                                              HI, HeI, HeII, krate24..26, Jmean1..3) ; real64 ksi(3,3) ; logtem0, logtem9, dlogtem ;
!                                              k1a..k6a(nratec)
!                                              out : real64 HI, HeI, HeII (ncell each) after ftteSolveRateEquations(n, runUVB)
!   dropin_check stellar <case.bin> <out.bin>  case: int32 n, ncell, nsrc, dustApproximation ; real64 box ; int32 level ;
!                                              real64 f(ncell,5) (HI, HeI, HeII, rho, abun2) ; per star int32 level, weight,
!                                              position(33) ; real64 a_smc(7,5), wavelength(1221), specificLuminosity(5,37,1221),
!                                              metallicity(5) ; int32 iSpectrum ; real64 coefSpectrum
!                                              out : real64 krate24..26, crate24..26 (ncell each) after ftteRunStellarTransfer
program dropin_check

  use definitions
  use ftte_uvb_transfer
  use ftte_rate_equations
  use ftte_stellar_transfer
  use dust

  implicit none
  integer :: n, ncell, ios, cursor, bi, bj, bk, runUVB
  integer, allocatable :: lev(:)
  real(kind=RealKind), allocatable :: f(:,:), outv(:,:)
  real(kind=RealKind) :: box, uvbIn(3), ksiIn(3,3)
  character(len=512) :: what, caseName, outName
  logical :: chem, stellarMode
  integer :: nsrc, is, iSpectrumIn, lvl, wgt, seq(33)
  real(kind=RealKind) :: coefSpectrumIn
  real(kind=RealKind), allocatable :: kout(:,:)
  type(starType), allocatable, target :: stars(:)

  call get_command_argument(1, what)
  call get_command_argument(2, caseName)
  call get_command_argument(3, outName)
  chem = trim(what) == 'chem'
  stellarMode = trim(what) == 'stellar'
  open(11, file=trim(caseName), access='stream', form='unformatted', status='old', iostat=ios)
  if (ios /= 0) stop 'dropin_check: cannot open case file'
  if (stellarMode) then
     read(11) n, ncell, nsrc, dustApproximation
     read(11) box
     allocate(lev(ncell), f(ncell,11), outv(ncell,3), kout(ncell,6), stars(nsrc))
     f = 0.
     read(11) lev
     read(11) f(:,1:5)
     do is = 1, nsrc
        read(11) lvl, wgt, seq
        stars(is)%level = lvl
        stars(is)%weight = wgt
        allocate(stars(is)%position(3*lvl+3))
        stars(is)%position = seq(1:3*lvl+3)
     enddo
     read(11) a_smc
     read(11) wavelength
     read(11) specificLuminosity
     read(11) metallicity
     read(11) iSpectrumIn
     read(11) coefSpectrumIn
  else if (chem) then
     read(11) n, ncell, runUVB
     read(11) box
     allocate(lev(ncell), f(ncell,11), outv(ncell,3))
     read(11) lev
     read(11) f
     read(11) ksiIn
     read(11) logtem0, logtem9, dlogtem
     read(11) k1a, k2a, k3a, k4a, k5a, k6a
     group1%ksi24 = ksiIn(1,1) ; group1%ksi25 = ksiIn(2,1) ; group1%ksi26 = ksiIn(3,1)
     group2%ksi24 = ksiIn(1,2) ; group2%ksi25 = ksiIn(2,2) ; group2%ksi26 = ksiIn(3,2)
     group3%ksi24 = ksiIn(1,3) ; group3%ksi25 = ksiIn(2,3) ; group3%ksi26 = ksiIn(3,3)
     uniformQuasar = 0. ; uniformStellar = 0.
     quasar%ksi24 = 0. ; quasar%ksi25 = 0. ; quasar%ksi26 = 0.
     stellar%ksi24 = 0. ; stellar%ksi25 = 0. ; stellar%ksi26 = 0.
     selfShieldingThreshold = 0.
  else
     read(11) n, ncell
     read(11) box, uvbIn
     allocate(lev(ncell), f(ncell,11), outv(ncell,3))
     f = 0.
     read(11) lev
     read(11) f(:,1:3)
     uvb1 = uvbIn(1) ; uvb2 = uvbIn(2) ; uvb3 = uvbIn(3)
  endif
  close(11)
  physicalBoxSize = box

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
  if (cursor /= ncell) stop 'dropin_check: level list does not describe a tree of ncell leaves'

  if (stellarMode) then
     call ftteRunStellarTransfer(n, nsrc, stars, iSpectrumIn, coefSpectrumIn)
  else if (chem) then
     call ftteSolveRateEquations(n, runUVB /= 0)
  else
     call ftteRunUVBTransfer(n)
  endif

  cursor = 0
  do bi = 1, n
     do bj = 1, n
        do bk = 1, n
           call harvest(baseGrid%cell(bi,bj,bk))
        enddo
     enddo
  enddo
  open(12, file=trim(outName), access='stream', form='unformatted', status='replace')
  if (stellarMode) then
     write(12) kout
  else
     write(12) outv
  endif
  close(12)
  write(*,*) 'dropin_check OK'

contains

  recursive subroutine growCell(c, level)
    type(zoneType), target :: c
    integer, intent(in) :: level
    integer :: a, b, d
    cursor = cursor + 1
    if (cursor > ncell) stop 'dropin_check: ran past the end of the level list'
    nullify(c%cell)
    c%level = int(level,1)
    if (lev(cursor) == level) then
       c%refined = .false.
       if (stellarMode) then
          c%HI = f(cursor,1) ; c%HeI = f(cursor,2) ; c%HeII = f(cursor,3) ; c%rho = f(cursor,4) ; c%abun2 = f(cursor,5)
          c%krate24 = 0. ; c%krate25 = 0. ; c%krate26 = 0. ; c%crate24 = 0. ; c%crate25 = 0. ; c%crate26 = 0.
       else if (chem) then
          c%rho = f(cursor,1) ; c%tgas = f(cursor,2)
          c%HI = f(cursor,3) ; c%HeI = f(cursor,4) ; c%HeII = f(cursor,5)
          c%krate24 = f(cursor,6) ; c%krate25 = f(cursor,7) ; c%krate26 = f(cursor,8)
          c%Jmean1 = f(cursor,9) ; c%Jmean2 = f(cursor,10) ; c%Jmean3 = f(cursor,11)
       else
          c%kappa1 = f(cursor,1) ; c%kappa2 = f(cursor,2) ; c%kappa3 = f(cursor,3)
          c%Jmean1 = 0. ; c%Jmean2 = 0. ; c%Jmean3 = 0.
       endif
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
       stop 'dropin_check: level list is not depth-first'
    endif
  end subroutine growCell

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
       if (stellarMode) then
          kout(cursor,1) = c%krate24 ; kout(cursor,2) = c%krate25 ; kout(cursor,3) = c%krate26
          kout(cursor,4) = c%crate24 ; kout(cursor,5) = c%crate25 ; kout(cursor,6) = c%crate26
       else if (chem) then
          outv(cursor,1) = c%HI ; outv(cursor,2) = c%HeI ; outv(cursor,3) = c%HeII
       else
          outv(cursor,1) = c%Jmean1 ; outv(cursor,2) = c%Jmean2 ; outv(cursor,3) = c%Jmean3
       endif
    endif
  end subroutine harvest

end program dropin_check
