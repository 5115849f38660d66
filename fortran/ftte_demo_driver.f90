! ftte_demo_driver.f90 -- a Fortran host in the shape of the reference driver's diffuse block
! (equiSources.f90:1372-1808) running on the GPU through ftte_binding, without needing the
! reference's modules: uniform n^3 grid, three frequency groups as in the reference, the
! reference's own 12 * 4**(nAngularLevel-1) direction loop.
!
!   ftte_demo_driver [n] [nAngularLevel]        (defaults 64, 2)
!
! Prints J at a few cells and two invariants: kappa = 0 gives J = uvb (sum of weights = 1), and
! J never exceeds the inflow.
program ftte_demo_driver

  use, intrinsic :: iso_c_binding
  use ftte_binding
  implicit none

  integer :: n, level, ndir, i, iarg
  integer(c_int64_t) :: ncell, c
  type(c_ptr) :: ctx
  integer(c_int32_t), allocatable :: lev(:)
  real(c_double), allocatable :: kappa(:,:), J(:,:), phi(:), theta(:), w(:)
  real(c_double) :: uvb(3), box, x
  character(len=32) :: arg

  n = 64
  level = 2
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg,*) n
  endif
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read(arg,*) level
  endif
  ncell = int(n, c_int64_t)**3
  ndir = 12 * 4**(level-1)
  box = 1.d0
  uvb = (/ 1.d-21, 4.d-22, 1.d-22 /)

  allocate(lev(ncell), kappa(ncell,3), J(ncell,3), phi(ndir), theta(ndir), w(ndir))
  lev = 0
  ! a smooth synthetic opacity field, tau per cell between ~0.02 and ~2 in group 1
  do c = 1, ncell
     x = dble(mod(c*2654435761_c_int64_t, 1000003_c_int64_t)) / 1000003.d0
     kappa(c,1) = dble(n) * (0.02d0 + 2.d0*x*x)
     kappa(c,2) = 0.3d0 * kappa(c,1)
     kappa(c,3) = 0.d0
  enddo

  ! the reference's direction loop header, equiSources.f90:1385-1391
  do i = 1, ndir
     iarg = ftte_pix2ang_nest(2**(level-1), int(i-1, c_int64_t), phi(i), theta(i))
     if (iarg /= FTTE_OK) stop 'pix2ang_nest failed'
     w(i) = 1.d0 / dble(ndir)
  enddo

  call ftteCheck(c_null_ptr, ftte_create(ctx, 1, c_null_ptr), 'ftte_create')
  call ftteCheck(ctx, ftte_set_grid(ctx, n, n, n, ncell, lev, box), 'ftte_set_grid')
  call ftteCheck(ctx, ftte_set_opacity(ctx, 3, kappa), 'ftte_set_opacity')
  call ftteCheck(ctx, ftte_diffuse_sweep(ctx, ndir, phi, theta, w, uvb, J), 'ftte_diffuse_sweep')

  write(*,'(a,i5,a,i5,a)') ' grid ', n, '^3, ', ndir, ' directions, 3 groups'
  write(*,'(a,3es14.6)') ' J(1,1,1)        =', J(1,1), J(1,2), J(1,3)
  write(*,'(a,3es14.6)') ' J(centre)       =', J(ncell/2 + n*n/2 + n/2, 1), J(ncell/2 + n*n/2 + n/2, 2), &
       J(ncell/2 + n*n/2 + n/2, 3)
  write(*,'(a,3es14.6)') ' max J / uvb     =', maxval(J(:,1))/uvb(1), maxval(J(:,2))/uvb(2), maxval(J(:,3))/uvb(3)
  if (maxval(abs(J(:,3)/uvb(3) - 1.d0)) > 1.d-14) stop 'FAIL: transparent group must return the inflow'
  if (maxval(J(:,1)) > uvb(1)*(1.d0+1.d-14)) stop 'FAIL: J exceeds the inflow'
  write(*,*) 'ftte_demo_driver OK'

  call ftteCheck(ctx, ftte_destroy(ctx), 'ftte_destroy')

end program ftte_demo_driver
