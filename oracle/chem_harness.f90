! chem_harness.f90 -- TEST INFRASTRUCTURE ONLY (never linked into the product).
!
! Drives the reference's own ionisation-equilibrium update, the consumer of J and of the point-source rates
! (SURVEY.md 8(f) row F1):
!   * solveRateEquations (+ the function `opposite`): contained procedures of the reference's main program
!     (equiSources.f90:3459-3677, :5044-5058), lifted at build time by line range into the module `chemExtract`
!     (oracle/Makefile; object and .mod in oracle/_ref/ only);
!   * coll_rates (coll_rates.f): compiled where it lies, called per temperature bin as calc_rates.f:324-337 does to fill
!     the rate-coefficient tables k1a..k6a (table bounds as equiSources.f90:174-176).
!
! usage: chem_harness <case.bin> <out.bin>
! case.bin (stream): int32 n, ncell, runUVB ; real64 box ; int32 level(ncell) ;
!   real64 rho, tgas, HI, HeI, HeII, krate24, krate25, krate26, Jmean1, Jmean2, Jmean3 (ncell each) ;
!   real64 ksiIn(3,3)  ((ksi24, ksi25, ksi26) of group1..3) ; real64 uniform(3) (4 pi (uniformQuasar quasar%ksi + ...)
!   per reaction, entered through quasar%ksi with uniformQuasar = 1/(4 pi)) ; real64 selfShieldingThreshold
! out.bin: real64 logtem0, logtem9, dlogtem ; real64 k1a..k6a (nratec each) ; real64 HI, HeI, HeII (ncell each)
!          (runUVB = 2: assignUvbRadiation instead, with uvb1..3 = uniform(1:3); the last three arrays are Jmean1..3)
program chem_harness

  use definitions
  use transportRoutinesModule
  use chemExtract

  implicit none

  integer :: n, ncell, runUVB, ios, cursor, bi, bj, bk, it
  double precision :: logttt, ttt
  integer, allocatable :: lev(:)
  real(kind=RealKind), allocatable :: f(:,:), outv(:,:)
  real(kind=RealKind) :: box, ksiIn(3,3), uni(3)
  character(len=512) :: caseName, outName
  logical :: uvb

  call get_command_argument(1, caseName)
  call get_command_argument(2, outName)
  open(11, file=trim(caseName), access='stream', form='unformatted', status='old', iostat=ios)
  if (ios /= 0) stop 'chem_harness: cannot open case file'
  read(11) n, ncell, runUVB
  read(11) box
  allocate(lev(ncell), f(ncell,11), outv(ncell,3))
  read(11) lev
  read(11) f
  read(11) ksiIn
  read(11) uni
  read(11) selfShieldingThreshold
  close(11)
  physicalBoxSize = box
  uvb = runUVB /= 0

  group1%ksi24 = ksiIn(1,1) ; group1%ksi25 = ksiIn(2,1) ; group1%ksi26 = ksiIn(3,1)
  group2%ksi24 = ksiIn(1,2) ; group2%ksi25 = ksiIn(2,2) ; group2%ksi26 = ksiIn(3,2)
  group3%ksi24 = ksiIn(1,3) ; group3%ksi25 = ksiIn(2,3) ; group3%ksi26 = ksiIn(3,3)
  ! the uniform background enters as 4 pi (uniformQuasar quasar%ksi + uniformStellar stellar%ksi): one term suffices
  uniformQuasar = 1.
  uniformStellar = 0.
  quasar%ksi24 = uni(1) ; quasar%ksi25 = uni(2) ; quasar%ksi26 = uni(3)
  stellar%ksi24 = 0. ; stellar%ksi25 = 0. ; stellar%ksi26 = 0.

  ! equiSources.f90:174-189
  logtem0 = log(temstart)
  logtem9 = log(temend)
  dlogtem = (log(temend) - log(temstart))/real(nratec-1)
  ! calc_rates.f cannot run as a whole here: after the rate coefficients it reads two cooling-rate data files that do not
  ! ship with the reference (HII-ktbetas.tab, cratesHe.res, calc_rates.f:397-412).  Its first loop (:324-337), which is all
  ! that fills k1a..k6a, is restated; coll_rates itself is the reference's object code.
  do it = 1, nratec
     logttt = log(temstart) + real(it-1)*dlogtem
     ttt = exp(logttt)
     call coll_rates(ttt, k1a(it), k2a(it), k3a(it), k4a(it), k5a(it), k6a(it), k7a(it), k8a(it), k9a(it), k10a(it), &
          k11a(it), k12a(it), k13a(it), k14a(it), k15a(it), k16a(it), k17a(it), k18a(it), k19a(it), recombinationType)
  enddo

  ! ---- tree from the leaf list (readCellArray.f90:154-187)
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
  if (cursor /= ncell) stop 'chem_harness: level list does not describe a tree of ncell leaves'

  ! ---- equiSources.f90:1811-1819
  icosmic = 0
  ncosmic = ncell
  if (runUVB == 2) then
     ! the optically thin alternative to the sweep, transportRoutinesModule.f90:1056-1093 (inflow = the case's uniform(1:3))
     uvb1 = uni(1) ; uvb2 = uni(2) ; uvb3 = uni(3)
     do bi = 1, n
        do bj = 1, n
           do bk = 1, n
              call assignUvbRadiation(baseGrid%cell(bi,bj,bk))
           enddo
        enddo
     enddo
  else
  do bi = 1, n
     do bj = 1, n
        do bk = 1, n
           call solveRateEquations(baseGrid%cell(bi,bj,bk), n, uvb)
        enddo
     enddo
  enddo
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
  write(12) logtem0, logtem9, dlogtem
  write(12) k1a, k2a, k3a, k4a, k5a, k6a
  write(12) outv
  close(12)

contains

  recursive subroutine growCell(c, level)
    type(zoneType), target :: c
    integer, intent(in) :: level
    integer :: a, b, d
    cursor = cursor + 1
    if (cursor > ncell) stop 'chem_harness: ran past the end of the level list'
    nullify(c%cell)
    c%level = int(level,1)
    if (lev(cursor) == level) then
       c%refined = .false.
       c%rho = f(cursor,1)
       c%tgas = f(cursor,2)
       c%HI = f(cursor,3)
       c%HeI = f(cursor,4)
       c%HeII = f(cursor,5)
       c%krate24 = f(cursor,6)
       c%krate25 = f(cursor,7)
       c%krate26 = f(cursor,8)
       c%Jmean1 = f(cursor,9)
       c%Jmean2 = f(cursor,10)
       c%Jmean3 = f(cursor,11)
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
       stop 'chem_harness: level list is not depth-first'
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
       if (runUVB == 2) then
          outv(cursor,1) = c%Jmean1 ; outv(cursor,2) = c%Jmean2 ; outv(cursor,3) = c%Jmean3
       else
       outv(cursor,1) = c%HI
       outv(cursor,2) = c%HeI
       outv(cursor,3) = c%HeII
       endif
    endif
  end subroutine harvest

end program chem_harness
