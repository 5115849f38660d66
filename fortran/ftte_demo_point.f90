! ftte_demo_point.f90 -- a Fortran host for the point-source path in the shape of the reference's star loop
! (equiSources.f90:1246-1329), through ftte_binding, without the reference's modules.
!
!   ftte_demo_point <case.bin> <rates.bin>
!
! case.bin (stream): int32 n, ncell, nsrc, dust ; real64 box ; int32 level(ncell) ;
!   real64 HI, HeI, HeII, rho, abun2 (ncell each) ; int64 srcCell(nsrc) (0-based) ; real64 ndot(nsrc) ;
!   real64 tables(11**4 * 6)  (reactionRate1..3, energyRate1..3)
! rates.bin: real64 rates(ncell,6) ; int32 highestPixelLevel
program ftte_demo_point

  use, intrinsic :: iso_c_binding
  use ftte_binding
  implicit none

  integer(c_int32_t) :: n, ncell32, nsrc, dust
  integer(c_int64_t) :: ncell
  integer(c_int) :: highest
  type(c_ptr) :: ctx
  integer(c_int32_t), allocatable :: lev(:)
  integer(c_int64_t), allocatable :: src(:)
  real(c_double), allocatable :: med(:,:), ndot(:), tables(:), rates(:,:)
  real(c_double) :: box
  character(len=512) :: caseName, outName
  integer :: ios

  call get_command_argument(1, caseName)
  call get_command_argument(2, outName)
  open(11, file=trim(caseName), access='stream', form='unformatted', status='old', iostat=ios)
  if (ios /= 0) stop 'ftte_demo_point: cannot open case file'
  read(11) n, ncell32, nsrc, dust
  read(11) box
  ncell = ncell32
  allocate(lev(ncell), med(ncell,5), src(nsrc), ndot(nsrc), tables(6*11**4), rates(ncell,6))
  read(11) lev
  read(11) med
  read(11) src
  read(11) ndot
  read(11) tables
  close(11)

  call ftteCheck(c_null_ptr, ftte_create(ctx, 1, c_null_ptr), 'ftte_create')
  call ftteCheck(ctx, ftte_set_grid(ctx, n, n, n, ncell, lev, box), 'ftte_set_grid')
  call ftteCheck(ctx, ftte_set_medium(ctx, med(:,1), med(:,2), med(:,3), med(:,4), med(:,5), dust), 'ftte_set_medium')
  call ftteCheck(ctx, ftte_set_rate_tables(ctx, tables), 'ftte_set_rate_tables')
  call ftteCheck(ctx, ftte_set_zero_rates(ctx), 'ftte_set_zero_rates')
  call ftteCheck(ctx, ftte_point_sources(ctx, nsrc, src, ndot, highest), 'ftte_point_sources')
  call ftteCheck(ctx, ftte_get_point_rates(ctx, rates), 'ftte_get_point_rates')

  open(12, file=trim(outName), access='stream', form='unformatted', status='replace')
  write(12) rates
  write(12) highest
  close(12)
  write(*,'(a,i4,a,i9,a,i5,a,i2)') ' grid ', n, '^3 base, ', ncell, ' cells, ', nsrc, ' sources, highestPixelLevel ', highest
  write(*,*) 'ftte_demo_point OK'
  call ftteCheck(ctx, ftte_destroy(ctx), 'ftte_destroy')

end program ftte_demo_point
