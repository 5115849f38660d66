! ingest_harness.f90 -- TEST INFRASTRUCTURE ONLY (never linked into the product).
!
! Drives the reference's own grid ingest (SURVEY.md 8(f) F4) on synthetic per-level cell lists:
!   * equiSources.f90:427-618 -- base grid size from the level-1 count, bounding box and physicalBoxSize, positions normalised
!     to the unit box, zeroed base grid, smoothing of the level-1 abundances, and the loop that places every listed cell
!     (placeCellProjectWithVelocity, :1870-1974) -- lifted BY LINE RANGE at build time (oracle/Makefile) into module
!     ingestExtract as subroutine referenceIngest, together with the writer of the cell array, writeCell (:4044-4079);
!   * what precedes them in the main program is the HDF4 read (:316-423): the lists come from a plain binary file instead.
!
! usage: ingest_harness <case.bin> <out.bin>
! case.bin (stream): int32 nlevels, kinematics, metals ; per level: int32 ncell ; real32 pos(ncell,3), lT, lnH, lx(ncell)
!                    [vel(ncell,3)] [abun(ncell,4)]
! out.bin: int32 nx, ncell ; real64 physicalBoxSize ; int32 level(ncell) ; real32 HI, HeI, HeII, temperature, density (ncell each,
!          as writeCell stores them) [velx, vely, velz] [abun2] ; then the same fields in real64 as the tree holds them:
!          HI, HeI, HeII, tgas, rho, velx, vely, velz, abun2
program ingest_harness

  use definitions
  use ingestExtract

  implicit none

  integer :: nlevels, kin, met, level, nc, ios, nxOut, i, j, k, cursor
  type(readLevelType), dimension(:), pointer :: lists
  real(kind=RealKind), allocatable :: f64(:,:)
  character(len=512) :: caseName, outName

  call get_command_argument(1, caseName)
  call get_command_argument(2, outName)
  open(11, file=trim(caseName), access='stream', form='unformatted', status='old', iostat=ios)
  if (ios /= 0) stop 'ingest_harness: cannot open case file'
  read(11) nlevels, kin, met
  readKinematics = kin /= 0
  readMetals = met /= 0
  allocate(lists(nlevels))
  do level = 1, nlevels
     read(11) nc
     lists(level)%ncell = nc
     allocate(lists(level)%pos(nc,3), lists(level)%lT(nc), lists(level)%lnH(nc), lists(level)%lx(nc))
     read(11) lists(level)%pos, lists(level)%lT, lists(level)%lnH, lists(level)%lx
     if (readKinematics) then
        allocate(lists(level)%vel(nc,3))
        read(11) lists(level)%vel
     endif
     if (readMetals) then
        allocate(lists(level)%abun(nc,4))
        read(11) lists(level)%abun
     endif
  enddo
  close(11)

  call referenceIngest(lists, nlevels, nxOut)

  ! the cell array as writeIonization builds it (equiSources.f90:4810-4843): count, allocate, writeCell over the base cells
  cursor = 0
  do i = 1, nxOut
     do j = 1, nxOut
        do k = 1, nxOut
           call countLeaves(baseGrid%cell(i,j,k))
        enddo
     enddo
  enddo
  allocate(cellArrayLevel(cursor), cellArrayHI(cursor), cellArrayHeI(cursor), cellArrayHeII(cursor), cellArrayTemp(cursor), &
       cellArrayDensity(cursor), cellArrayVelx(cursor), cellArrayVely(cursor), cellArrayVelz(cursor), cellArrayAbun2(cursor))
  allocate(f64(cursor,9))
  icosmic = 0
  do i = 1, nxOut
     do j = 1, nxOut
        do k = 1, nxOut
           call writeCell(baseGrid%cell(i,j,k), 0)
        enddo
     enddo
  enddo
  if (icosmic /= cursor) stop 'ingest_harness: writeCell and the leaf count disagree'
  cursor = 0
  do i = 1, nxOut
     do j = 1, nxOut
        do k = 1, nxOut
           call harvest(baseGrid%cell(i,j,k))
        enddo
     enddo
  enddo

  open(12, file=trim(outName), access='stream', form='unformatted', status='replace')
  write(12) nxOut, icosmic
  write(12) physicalBoxSize
  write(12) cellArrayLevel
  write(12) cellArrayHI, cellArrayHeI, cellArrayHeII, cellArrayTemp, cellArrayDensity
  if (readKinematics) write(12) cellArrayVelx, cellArrayVely, cellArrayVelz
  if (readMetals) write(12) cellArrayAbun2
  write(12) f64
  close(12)

contains

  recursive subroutine countLeaves(c)
    type(zoneType) :: c
    integer :: a, b, d
    if (c%refined) then
       do a = 1, 2
          do b = 1, 2
             do d = 1, 2
                call countLeaves(c%cell(a,b,d))
             enddo
          enddo
       enddo
    else
       cursor = cursor + 1
    endif
  end subroutine countLeaves

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
       f64(cursor,:) = (/ c%HI, c%HeI, c%HeII, c%tgas, c%rho, c%velx, c%vely, c%velz, c%abun2 /)
    endif
  end subroutine harvest

end program ingest_harness
