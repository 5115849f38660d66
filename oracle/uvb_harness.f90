! uvb_harness.f90 -- TEST INFRASTRUCTURE ONLY.  Calls the reference's uvbBetaTable (uvbBetaTable.f90, compiled where it
! lies) as the main program does (equiSources.f90:253) and writes the nine beta, ksi and gamma values it leaves in group1..3.
! usage: uvb_harness <alpha1> <alpha2> <alpha3> <out.bin>
! out.bin (after the group tables, from uniformTable(alpha1, alpha2)): (ksi24, ksi25, ksi26) x (quasar, stellar) ; (gammaHI, gammaHeI, gammaHeII) x (quasar, stellar)
! out.bin: real64 (beta24, beta25, beta26) x group1..3 ; (ksi24, ksi25, ksi26) x group1..3 ; (gammaHI, gammaHeI, gammaHeII) x group1..3
program uvb_harness
  use definitions
  implicit none
  real(kind=RealKind) :: alpha(3)
  character(len=512) :: arg
  integer :: q, it, rtype
  double precision :: ttt, kk(19), kout(6,64,2), tout(64)
  do q = 1, 3
     call get_command_argument(q, arg)
     read(arg,*) alpha(q)
  enddo
  call get_command_argument(4, arg)
  call uvbBetaTable(nfbins, frequencyBinWidth, alpha)
  ! the uniform background's two components with the first two slopes (uniformTable.f90, called at equiSources.f90:192)
  call uniformTable(nfbins, frequencyBinWidth, alpha(1), alpha(2))
  ! coll_rates (coll_rates.f) at 64 temperatures between 1 and 1e8 K, recombination case A (1) and case B (2)
  do rtype = 1, 2
     do it = 1, 64
        ttt = 10.d0**(8.d0*dble(it-1)/63.d0)
        call coll_rates(ttt, kk(1), kk(2), kk(3), kk(4), kk(5), kk(6), kk(7), kk(8), kk(9), kk(10), kk(11), kk(12), kk(13), &
             kk(14), kk(15), kk(16), kk(17), kk(18), kk(19), rtype)
        kout(:,it,rtype) = kk(1:6)
        tout(it) = ttt
     enddo
  enddo
  open(12, file=trim(arg), access='stream', form='unformatted', status='replace')
  write(12) group1%beta24, group1%beta25, group1%beta26, group2%beta24, group2%beta25, group2%beta26, &
       group3%beta24, group3%beta25, group3%beta26
  write(12) group1%ksi24, group1%ksi25, group1%ksi26, group2%ksi24, group2%ksi25, group2%ksi26, &
       group3%ksi24, group3%ksi25, group3%ksi26
  write(12) group1%gammaHI, group1%gammaHeI, group1%gammaHeII, group2%gammaHI, group2%gammaHeI, group2%gammaHeII, &
       group3%gammaHI, group3%gammaHeI, group3%gammaHeII
  write(12) quasar%ksi24, quasar%ksi25, quasar%ksi26, stellar%ksi24, stellar%ksi25, stellar%ksi26
  write(12) quasar%gammaHI, quasar%gammaHeI, quasar%gammaHeII, stellar%gammaHI, stellar%gammaHeI, stellar%gammaHeII
  write(12) kout
  write(12) tout
  close(12)
end program uvb_harness
