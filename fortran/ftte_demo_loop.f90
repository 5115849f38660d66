! ftte_demo_loop.f90 -- a Fortran host for the reference's whole outer iteration (equiSources.f90:1230-1843) on the GPU, through
! ftte_binding and without the reference's modules: per iteration
!     setZeroRates -> star loop (ftte_point_sources) -> computeOpacities (ftte_compute_opacities) ->
!     diffuse sweep over 12*4**(L-1) directions (ftte_diffuse_sweep) -> solveRateEquations (ftte_solve_rate_equations)
! with the group cross-sections and photo-rate coefficients of uvbBetaTable (ftte_uvb_beta_table).
!
!   ftte_demo_loop <case.bin> <out.bin>
!
! case.bin (stream): int32 n, ncell, nsrc, niter, angularLevel, nratec ; real64 box, alpha(3), uvb(3) ; int32 level(ncell) ;
!   real64 rho, tgas, HI, HeI, HeII (ncell each) ; int64 srcCell(nsrc) (0-based) ; real64 ndot(nsrc) ;
!   real64 tables(6*11**4) ; real64 logtem0, logtem9, dlogtem ; real64 k(nratec,6)
! out.bin: real64 HI, HeI, HeII (ncell each) after the last iteration ; real64 J(ncell,3) of the last sweep
program ftte_demo_loop

  use, intrinsic :: iso_c_binding
  use ftte_binding
  implicit none

  integer(c_int32_t) :: n, ncell32, nsrc, niter, angularLevel, nratec
  integer(c_int64_t) :: ncell, ipix
  integer(c_int) :: highest, ndir, i, it
  type(c_ptr) :: ctx
  integer(c_int32_t), allocatable :: lev(:)
  integer(c_int64_t), allocatable :: src(:)
  real(c_double), allocatable :: gas(:,:), ndot(:), tables(:), k(:,:), J(:,:), phi(:), theta(:), w(:), species(:,:)
  real(c_double) :: box, alpha(3), uvb(3), logtem(3), beta(3,3), ksi(3,3), gam(3,3), uniform(3), change
  real(c_double), allocatable :: kown(:,:)
  real(c_double) :: logtemOwn(3)
  character(len=512) :: caseName, outName
  integer :: ios

  call get_command_argument(1, caseName)
  call get_command_argument(2, outName)
  open(11, file=trim(caseName), access='stream', form='unformatted', status='old', iostat=ios)
  if (ios /= 0) stop 'ftte_demo_loop: cannot open case file'
  read(11) n, ncell32, nsrc, niter, angularLevel, nratec
  read(11) box, alpha, uvb
  ncell = ncell32
  allocate(lev(ncell), gas(ncell,5), src(nsrc), ndot(nsrc), tables(6*11**4), k(nratec,6), J(ncell,3), species(ncell,3))
  read(11) lev
  read(11) gas
  read(11) src
  read(11) ndot
  read(11) tables
  read(11) logtem
  read(11) k
  close(11)

  ! the direction loop header of the reference, equiSources.f90:1385-1391
  ndir = 12 * 4**(angularLevel-1)
  allocate(phi(ndir), theta(ndir), w(ndir))
  do i = 1, ndir
     ipix = i - 1
     if (ftte_pix2ang_nest(2**(angularLevel-1), ipix, phi(i), theta(i)) /= FTTE_OK) stop 'pix2ang_nest failed'
  enddo
  w = 1./float(ndir)
  uniform = 0.d0

  call ftteCheck(c_null_ptr, ftte_create(ctx, 1, c_null_ptr), 'ftte_create')
  ! uvbBetaTable with the reference's nfbins and frequencyBinWidth (a default-real 0.02)
  call ftteCheck(ctx, ftte_uvb_beta_table(400, real(0.02, c_double), alpha, beta, ksi, gam), 'ftte_uvb_beta_table')
  ! the rate-coefficient tables the reference's driver builds with calc_rates/coll_rates (case B, 1 K .. 1e8 K): when the case
  ! file carries the reference's own, the library's must be the same numbers
  allocate(kown(nratec,6))
  call ftteCheck(ctx, ftte_rate_coefficient_tables(nratec, 1.d0, real(1.e8, c_double), 2, kown, logtemOwn(1), logtemOwn(2), &
       logtemOwn(3)), 'ftte_rate_coefficient_tables')
  if (any(kown /= k) .or. any(logtemOwn /= logtem)) stop 'ftte_demo_loop: rate-coefficient tables differ from the case file'
  call ftteCheck(ctx, ftte_set_grid(ctx, n, n, n, ncell, lev, box), 'ftte_set_grid')
  call ftteCheck(ctx, ftte_set_rate_tables(ctx, tables), 'ftte_set_rate_tables')
  call ftteCheck(ctx, ftte_set_rate_coefficients(ctx, nratec, logtem(1), logtem(2), logtem(3), k(:,1), k(:,2), k(:,3), k(:,4), &
       k(:,5), k(:,6)), 'ftte_set_rate_coefficients')
  call ftteCheck(ctx, ftte_set_medium(ctx, gas(:,3), gas(:,4), gas(:,5), gas(:,1), gas(:,1), 0), 'ftte_set_medium')
  call ftteCheck(ctx, ftte_set_temperature(ctx, gas(:,2)), 'ftte_set_temperature')

  do it = 1, niter
     call ftteCheck(ctx, ftte_set_zero_rates(ctx), 'ftte_set_zero_rates')
     call ftteCheck(ctx, ftte_point_sources(ctx, nsrc, src, ndot, highest), 'ftte_point_sources')
     call ftteCheck(ctx, ftte_compute_opacities(ctx, 3, beta), 'ftte_compute_opacities')
     call ftteCheck(ctx, ftte_diffuse_sweep(ctx, ndir, phi, theta, w, uvb, J), 'ftte_diffuse_sweep')
     call ftteCheck(ctx, ftte_solve_rate_equations(ctx, 1, J, ksi, uniform, 0.d0, 1, change), 'ftte_solve_rate_equations')
     write(*,'(a,i3,a,es12.4,a,i2)') ' iteration ', it, ': largest change of a species fraction ', change, &
          ', highestPixelLevel ', highest
  enddo
  call ftteCheck(ctx, ftte_get_medium(ctx, species(:,1), species(:,2), species(:,3)), 'ftte_get_medium')

  open(12, file=trim(outName), access='stream', form='unformatted', status='replace')
  write(12) species
  write(12) J
  close(12)
  write(*,*) 'ftte_demo_loop OK'
  call ftteCheck(ctx, ftte_destroy(ctx), 'ftte_destroy')

end program ftte_demo_loop
