! dat_harness -- TEST INFRASTRUCTURE ONLY.  The reference's converter hdf42bin.f90 with its HDF4 input replaced by a
! case file: the module (lines 1-75), the cell-centre loop and scaling (162-192), the write statements (208-218) and
! computeCellCoordinates (222-269) are the reference's own lines, spliced around this program head by oracle/Makefile at
! build time (temporary file outside the repository).
!
! usage: dat_harness <case.bin> <output directory with trailing slash>
! case.bin (stream): int32 n, ncell ; real64 physicalBoxSize [cm] ; int32 level(ncell) ; real32 HI, HeI, HeII, T, density
program dat_harness

  use localDefinitions

  implicit none
  integer :: i, j, k, nx, ny, nz, ios
  real(kind=RealKind) :: physicalBoxSize, xa, xb, ya, yb, za, zb, xpos, ypos, zpos
  integer, dimension(0:30) :: numberCells
  character(60) :: dirname, basename
  character(len=512) :: caseName

  call get_command_argument(1, caseName)
  call get_command_argument(2, dirname)
  basename = 'cellArray'
  open(11, file=trim(caseName), access='stream', form='unformatted', status='old', iostat=ios)
  if (ios /= 0) stop 'dat_harness: cannot open case file'
  read(11) nx, ncosmic
  read(11) physicalBoxSize
  ny = nx
  nz = nx
  allocate(cellArrayLevel(ncosmic), cellArrayXpos(ncosmic), cellArrayYpos(ncosmic), cellArrayZpos(ncosmic))
  allocate(cellArrayHI(ncosmic), cellArrayHeI(ncosmic), cellArrayHeII(ncosmic), cellArrayTemp(ncosmic), cellArrayDensity(ncosmic))
  read(11) cellArrayLevel
  read(11) cellArrayHI, cellArrayHeI, cellArrayHeII, cellArrayTemp, cellArrayDensity
  close(11)

