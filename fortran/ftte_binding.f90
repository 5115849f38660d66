! ftte_binding.f90 -- ISO_C_BINDING view of include/ftte.h for the Fortran host.
!
! The reference is a Fortran program; its diffuse-transfer block (equiSources.f90:1372-1808) is
! replaced by calls into libftte.so through these interfaces.  Nothing here computes: it is the
! thin shim the north star asks for.  Status codes are those of ftte_status in ftte.h; the
! wrapper `ftteCheck` reproduces the reference's behaviour on inconsistency (`write; stop`).
module ftte_binding

  use, intrinsic :: iso_c_binding
  implicit none

  integer(c_int), parameter :: FTTE_OK = 0

  interface

     integer(c_int) function ftte_create(ctx, ndev, dev_ids) bind(C, name='ftte_create')
       import :: c_ptr, c_int
       type(c_ptr), intent(out) :: ctx
       integer(c_int), value :: ndev
       type(c_ptr), value :: dev_ids            ! const int*, may be c_null_ptr
     end function ftte_create

     integer(c_int) function ftte_destroy(ctx) bind(C, name='ftte_destroy')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function ftte_destroy

     type(c_ptr) function ftte_last_error(ctx) bind(C, name='ftte_last_error')
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function ftte_last_error

     type(c_ptr) function ftte_multi_info(ctx) bind(C, name='ftte_multi_info')
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function ftte_multi_info

     integer(c_int) function ftte_set_grid(ctx, nx, ny, nz, ncell, level, box_cm) bind(C, name='ftte_set_grid')
       import :: c_ptr, c_int, c_int64_t, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nx, ny, nz
       integer(c_int64_t), value :: ncell
       integer(c_int32_t), intent(in) :: level(*)
       real(c_double), value :: box_cm
     end function ftte_set_grid

     integer(c_int) function ftte_set_opacity(ctx, nnu, kappa) bind(C, name='ftte_set_opacity')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nnu
       real(c_double), intent(in) :: kappa(*)   ! (ncell, nnu) in Fortran order == [nnu][ncell] in C
     end function ftte_set_opacity

     integer(c_int) function ftte_set_species(ctx, nnu, HI, HeI, HeII, beta) bind(C, name='ftte_set_species')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nnu
       real(c_double), intent(in) :: HI(*), HeI(*), HeII(*)
       real(c_double), intent(in) :: beta(*)    ! (nnu, 3) in Fortran order == [3][nnu] in C
     end function ftte_set_species

     integer(c_int) function ftte_set_emissivity(ctx, eta) bind(C, name='ftte_set_emissivity')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       type(c_ptr), value :: eta                ! const double* (ncell, nnu); c_null_ptr: the reference's zero emissivity
     end function ftte_set_emissivity

     integer(c_int) function ftte_set_source_function(ctx, S) bind(C, name='ftte_set_source_function')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
       type(c_ptr), value :: S                  ! const double* (ncell, nnu), or c_null_ptr: emission off
     end function ftte_set_source_function

     integer(c_int) function ftte_diffuse_sweep(ctx, ndir, phi, theta, w, uvb, J) bind(C, name='ftte_diffuse_sweep')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: ndir
       real(c_double), intent(in) :: phi(*), theta(*), w(*), uvb(*)
       real(c_double), intent(out) :: J(*)      ! (ncell, nnu)
     end function ftte_diffuse_sweep

     ! ftte_set_opacity + ftte_diffuse_sweep in one call; on a uniform grid the groups cross PCIe and are swept in overlapping lanes
     integer(c_int) function ftte_diffuse_iteration(ctx, nnu, kappa, ndir, phi, theta, w, uvb, J) bind(C, name='ftte_diffuse_iteration')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nnu, ndir
       real(c_double), intent(in) :: kappa(*)   ! (ncell, nnu)
       real(c_double), intent(in) :: phi(*), theta(*), w(*), uvb(*)
       real(c_double), intent(out) :: J(*)      ! (ncell, nnu)
     end function ftte_diffuse_iteration

     ! pins a host array the caller keeps (kappa, J): moved by DMA without a staging copy while it stays registered
     integer(c_int) function ftte_host_register(ctx, ptr, bytes) bind(C, name='ftte_host_register')
       import :: c_ptr, c_int, c_size_t
       type(c_ptr), value :: ctx, ptr
       integer(c_size_t), value :: bytes
     end function ftte_host_register

     integer(c_int) function ftte_host_unregister(ctx, ptr) bind(C, name='ftte_host_unregister')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx, ptr
     end function ftte_host_unregister

     ! "grid_builds", "plan_builds", "forest_builds" (null-terminated): how often the host-side builds ran
     integer(c_long_long) function ftte_counter(ctx, name) bind(C, name='ftte_counter')
       import :: c_ptr, c_long_long, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: name(*)
     end function ftte_counter

     integer(c_int) function ftte_set_option(ctx, key, val) bind(C, name='ftte_set_option')
       import :: c_ptr, c_int, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: key(*)
       integer(c_int), value :: val
     end function ftte_set_option

     integer(c_int) function ftte_pix2ang_nest(nside, ipix, phi, theta) bind(C, name='ftte_pix2ang_nest')
       import :: c_int, c_int64_t, c_double
       integer(c_int), value :: nside
       integer(c_int64_t), value :: ipix
       real(c_double), intent(out) :: phi, theta
     end function ftte_pix2ang_nest

     integer(c_int) function ftte_fold_direction(phi_large, theta_large, phi, theta, izone) bind(C, name='ftte_fold_direction')
       import :: c_int, c_double
       real(c_double), value :: phi_large, theta_large
       real(c_double), intent(out) :: phi, theta
       integer(c_int), intent(out) :: izone
     end function ftte_fold_direction

     integer(c_int) function ftte_rotate_indices(i, j, k, nx, ny, nz, izone, icell, jcell, kcell) &
          bind(C, name='ftte_rotate_indices')
       import :: c_int
       integer(c_int), value :: i, j, k, nx, ny, nz, izone
       integer(c_int), intent(out) :: icell, jcell, kcell
     end function ftte_rotate_indices

     ! ---- point sources (the runStellarTransfer block, equiSources.f90:1256-1370)

     integer(c_int) function ftte_stellar_beta_table(ctx, a_smc, nwave, wavelength_cm, nspectrum, nmetal, &
          specific_luminosity, iSpectrum, coefSpectrum, iMetal, coefMetal, total_integral) &
          bind(C, name='ftte_stellar_beta_table')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: a_smc(7,5)               ! module dust
       integer(c_int), value :: nwave, nspectrum, nmetal
       real(c_double), intent(in) :: wavelength_cm(*)         ! wavelength(nWavelengths)
       real(c_double), intent(in) :: specific_luminosity(*)   ! specificLuminosity(nMetallicity,nSpectra,nWavelengths), as is
       integer(c_int), value :: iSpectrum, iMetal
       real(c_double), value :: coefSpectrum, coefMetal
       real(c_double), intent(out) :: total_integral
     end function ftte_stellar_beta_table

     integer(c_int) function ftte_set_rate_tables(ctx, tables) bind(C, name='ftte_set_rate_tables')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: tables(*)   ! reactionRate1..3, energyRate1..3, each (0:10,0:10,0:10,0:10)
     end function ftte_set_rate_tables

     integer(c_int) function ftte_get_rate_tables(ctx, tables) bind(C, name='ftte_get_rate_tables')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: tables(*)
     end function ftte_get_rate_tables

     integer(c_int) function ftte_get_rates_hydrogen_helium(ctx, dust_approximation, nsample, tau, rates) &
          bind(C, name='ftte_get_rates_hydrogen_helium')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: dust_approximation, nsample
       real(c_double), intent(in) :: tau(*)      ! (4, nsample)
       real(c_double), intent(out) :: rates(*)   ! (2, 3, nsample): numberRate, heatingRate per reaction
     end function ftte_get_rates_hydrogen_helium

     integer(c_int) function ftte_set_medium(ctx, HI, HeI, HeII, rho, abun2, dust_approximation) &
          bind(C, name='ftte_set_medium')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: HI(*), HeI(*), HeII(*), rho(*), abun2(*)
       integer(c_int), value :: dust_approximation
     end function ftte_set_medium

     integer(c_int) function ftte_set_zero_rates(ctx) bind(C, name='ftte_set_zero_rates')
       import :: c_ptr, c_int
       type(c_ptr), value :: ctx
     end function ftte_set_zero_rates

     integer(c_int) function ftte_locate_cell(ctx, level, position, cell) bind(C, name='ftte_locate_cell')
       import :: c_ptr, c_int, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: level
       integer(c_int32_t), intent(in) :: position(*)   ! starType%position(1:3*level+3)
       integer(c_int64_t), intent(out) :: cell         ! 0-based cell-array index
     end function ftte_locate_cell

     integer(c_int) function ftte_point_sources(ctx, nsrc, src_cell, src_ndot, highest_pixel_level) &
          bind(C, name='ftte_point_sources')
       import :: c_ptr, c_int, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nsrc
       integer(c_int64_t), intent(in) :: src_cell(*)
       real(c_double), intent(in) :: src_ndot(*)
       integer(c_int), intent(out) :: highest_pixel_level
     end function ftte_point_sources

     ! escape bookkeeping of the last ftte_point_sources call (equiSources.f90:3198-3233, 1342-1348); arrays as Fortran
     ! (7,nsrc), (7,nsrc), (nsrc), (300,nsrc), (7,nsrc); pass c_null_ptr-associated dummies by using the _opt variant if unwanted
     integer(c_int) function ftte_point_escape(ctx, nsrc, remaining, boundary, dust, spectrum, fraction) bind(C, name='ftte_point_escape')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nsrc
       real(c_double), intent(out) :: remaining(*), boundary(*), dust(*), spectrum(*), fraction(*)
     end function ftte_point_escape

     integer(c_int) function ftte_set_output_sigma(ctx, sigma) bind(C, name='ftte_set_output_sigma')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: sigma(*)
     end function ftte_set_output_sigma

     integer(c_int) function ftte_get_point_rates(ctx, rates) bind(C, name='ftte_get_point_rates')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: rates(*)   ! (ncell, 6): krate24, krate25, krate26, crate24, crate25, crate26
     end function ftte_get_point_rates

     integer(c_int) function ftte_rmax(rmax30) bind(C, name='ftte_rmax')
       import :: c_int, c_double
       real(c_double), intent(out) :: rmax30(30)
     end function ftte_rmax

     real(c_double) function ftte_dust_cross_section(lambda_micron, a_smc) bind(C, name='ftte_dust_cross_section')
       import :: c_double
       real(c_double), value :: lambda_micron
       real(c_double), intent(in) :: a_smc(7,5)
     end function ftte_dust_cross_section

     ! ---- ionisation equilibrium (solveRateEquations, equiSources.f90:3459-3677)

     integer(c_int) function ftte_set_rate_coefficients(ctx, nratec, logtem0, logtem9, dlogtem, k1a, k2a, k3a, k4a, k5a, k6a) &
          bind(C, name='ftte_set_rate_coefficients')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nratec
       real(c_double), value :: logtem0, logtem9, dlogtem
       real(c_double), intent(in) :: k1a(*), k2a(*), k3a(*), k4a(*), k5a(*), k6a(*)
     end function ftte_set_rate_coefficients

     integer(c_int) function ftte_set_temperature(ctx, tgas) bind(C, name='ftte_set_temperature')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: tgas(*)
     end function ftte_set_temperature

     integer(c_int) function ftte_solve_rate_equations(ctx, run_uvb_transfer, J, ksi, uniform, self_shielding_threshold, &
          use_point_rates, max_change) bind(C, name='ftte_solve_rate_equations')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: run_uvb_transfer, use_point_rates
       real(c_double), intent(in) :: J(*)          ! (ncell, 3): Jmean1..3
       real(c_double), intent(in) :: ksi(3,3)      ! (ksi24, ksi25, ksi26) x (group1, group2, group3)
       real(c_double), intent(in) :: uniform(3)
       real(c_double), value :: self_shielding_threshold
       real(c_double), intent(out) :: max_change
     end function ftte_solve_rate_equations

     integer(c_int) function ftte_get_medium(ctx, HI, HeI, HeII) bind(C, name='ftte_get_medium')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: HI(*), HeI(*), HeII(*)
     end function ftte_get_medium

     integer(c_int) function ftte_compute_opacities(ctx, nnu, beta) bind(C, name='ftte_compute_opacities')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nnu
       real(c_double), intent(in) :: beta(*)       ! (nnu, 3)
     end function ftte_compute_opacities

     integer(c_int) function ftte_set_point_rates(ctx, rates) bind(C, name='ftte_set_point_rates')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: rates(*)      ! (ncell, 6)
     end function ftte_set_point_rates

     integer(c_int) function ftte_uvb_beta_table(nfreq, freqdel, alpha, beta, ksi, gamma) bind(C, name='ftte_uvb_beta_table')
       import :: c_int, c_double
       integer(c_int), value :: nfreq
       real(c_double), value :: freqdel
       real(c_double), intent(in) :: alpha(3)
       real(c_double), intent(out) :: beta(3,3)    ! (group, species HI/HeI/HeII): the (nnu, 3) matrix of ftte_set_species
       real(c_double), intent(out) :: ksi(3,3)     ! (ksi24/25/26, group)
       real(c_double), intent(out) :: gamma(3,3)   ! (gammaHI/HeI/HeII, group)
     end function ftte_uvb_beta_table

     integer(c_int) function ftte_assign_uvb_radiation(ctx, nnu, uvb, self_shielding_threshold, J) &
          bind(C, name='ftte_assign_uvb_radiation')
       import :: c_ptr, c_int, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nnu
       real(c_double), intent(in) :: uvb(*)
       real(c_double), value :: self_shielding_threshold
       real(c_double), intent(out) :: J(*)         ! (ncell, nnu)
     end function ftte_assign_uvb_radiation

     integer(c_int) function ftte_uniform_table(nfreq, freqdel, alpha_quasar, alpha_stellar, ksi, gamma) &
          bind(C, name='ftte_uniform_table')
       import :: c_int, c_double
       integer(c_int), value :: nfreq
       real(c_double), value :: freqdel, alpha_quasar, alpha_stellar
       real(c_double), intent(out) :: ksi(3,2), gamma(3,2)   ! (24/25/26 or HI/HeI/HeII, quasar/stellar)
     end function ftte_uniform_table

     integer(c_int) function ftte_rate_coefficient_tables(nratec, temstart, temend, recombination_type, k, logtem0, logtem9, dlogtem) &
          bind(C, name='ftte_rate_coefficient_tables')
       import :: c_int, c_double
       integer(c_int), value :: nratec, recombination_type
       real(c_double), value :: temstart, temend
       real(c_double), intent(out) :: k(*)         ! (nratec, 6): k1a..k6a
       real(c_double), intent(out) :: logtem0, logtem9, dlogtem
     end function ftte_rate_coefficient_tables

  end interface

contains

  ! the reference's error convention: print and stop (e.g. equiSources.f90:1412, transportRoutinesModule.f90:33-36)
  subroutine ftteCheck(ctx, status, where)
    type(c_ptr), intent(in) :: ctx
    integer(c_int), intent(in) :: status
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: n
    if (status == FTTE_OK) return
    p = ftte_last_error(ctx)
    if (c_associated(p)) then
       call c_f_pointer(p, msg, [1024])
       n = 0
       do while (n < 1024)
          if (msg(n+1) == c_null_char) exit
          n = n + 1
       enddo
       write(*,*) 'ftte error in ', where, ': status', status, ' ', msg(1:n)
    else
       write(*,*) 'ftte error in ', where, ': status', status
    endif
    stop 1
  end subroutine ftteCheck

end module ftte_binding
