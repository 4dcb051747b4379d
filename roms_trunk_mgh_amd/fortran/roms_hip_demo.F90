!=======================================================================
!  roms_hip_demo -- a Fortran host for libroms_hip.so, used by
!  tests/test_gpu_fortran_host.py: the drop-in boundary exercised from the
!  language the reference's driver is written in.  It is NOT part of ROMS:
!  it reads one tile's state (the registered fields, the bounds and the
!  parameter block as written by tests/test_gpu_fortran_host.py), registers
!  the arrays through c_loc as INTEGRATION.md describes for mod_ocean /
!  mod_grid / mod_coupling, issues the calls of main3d in main3d's order
!  (ROMS/Nonlinear/main3d.F:189-191, :307-314, :467-475, :489, :563,
!  :592-700, :736, :762, :789, :814) through roms_hip_mod, and writes the
!  prognostic fields back.
!
!    roms_hip_demo <state.bin> <result.bin> <nsteps>
!=======================================================================
PROGRAM roms_hip_demo
  USE, INTRINSIC :: iso_c_binding
  USE roms_hip_mod
  IMPLICIT NONE
  INTEGER, PARAMETER :: MAXF = 256
  TYPE fld
    REAL(c_double), POINTER :: a(:) => NULL()
  END TYPE fld
  TYPE(fld) :: F(0:MAXF-1)
  INTEGER(c_int8_t), ALLOCATABLE, TARGET :: bimg(:), pimg(:)
  INTEGER(c_long) :: nb, np, nf, id, cnt, ns, nq, nt
  !  SOURCES(ng) as set_data leaves it (mod_sources.F), when the state file carries one
  INTEGER(c_int), ALLOCATABLE :: Isrc(:), Jsrc(:), ltr(:)
  REAL(c_double), ALLOCATABLE :: Dsrc(:), Qbar(:), Qsrc(:), Tsrc(:)
  INTEGER :: ios
  INTEGER :: nsteps, istep, q, iic, ntstart, exit_flag
  INTEGER(c_int) :: indx1, rc
  TYPE(roms_step_idx_t) :: s
  REAL(c_double) :: d12(12)
  CHARACTER(len=512) :: fin, fout, arg
  INTEGER(c_int), PARAMETER :: outids(6) = (/ FID_zeta, FID_ubar, FID_vbar, FID_u, FID_v, FID_t /)

  CALL get_command_argument (1, fin)
  CALL get_command_argument (2, fout)
  CALL get_command_argument (3, arg)
  READ (arg, *) nsteps
  exit_flag = 0

  OPEN (10, FILE=TRIM(fin), ACCESS='stream', FORM='unformatted', STATUS='old')
  READ (10) nb
  ALLOCATE (bimg(nb))
  READ (10) bimg
  READ (10) np
  ALLOCATE (pimg(np))
  READ (10) pimg
  READ (10) nf
  DO q = 1, INT(nf)
    READ (10) id, cnt
    ALLOCATE (F(id)%a(cnt))
    READ (10) F(id)%a
  END DO
  ns = 0
  READ (10, IOSTAT=ios) ns, nq, nt             ! optional: Nsrc, Nsrc*N, NT
  IF (ios /= 0) ns = 0
  IF (ns > 0) THEN
    ALLOCATE (Isrc(ns), Jsrc(ns), Dsrc(ns), Qbar(ns), Qsrc(nq), Tsrc(nq*nt), ltr(nt))
    READ (10) Isrc, Jsrc, Dsrc, Qbar, Qsrc, Tsrc, ltr
  END IF
  CLOSE (10)

  CALL check (roms_hip_init (0_c_int, 1_c_int, 1_c_int, 0_c_int, c_null_ptr), 'init')
  CALL check (roms_hip_set_bounds (c_loc(bimg)), 'set_bounds')
  CALL check (roms_hip_set_params (c_loc(pimg)), 'set_params')
  DO q = 0, MAXF-1
    IF (ASSOCIATED(F(q)%a)) THEN
      CALL check (roms_hip_register_field (INT(q, c_int), c_loc(F(q)%a(1)), INT(SIZE(F(q)%a), c_long)), 'register_field')
    END IF
  END DO
  CALL check (roms_hip_sync_all_to_device (), 'sync_all_to_device')
  !  point sources (LuvSrc / LwSrc): the table, once (a steady river; a driver calls this after every set_data)
  IF (ns > 0) THEN
    CALL check (roms_hip_set_sources (INT(ns, c_int), Isrc, Jsrc, Dsrc, Qbar, Qsrc, Tsrc, ltr), 'set_sources')
  END IF

  !  initial.F:337-571, the part on the path
  ntstart = 1
  s = roms_hip_make_idx (ntstart, ntstart, 1, 2, 1, 1, 1, 1, 1, .FALSE.)
  CALL check (roms_hip_set_depth (s), 'set_depth')
  CALL check (roms_hip_set_massflux (s), 'set_massflux')
  CALL check (roms_hip_omega (s), 'omega')
  CALL check (roms_hip_rho_eos (s), 'rho_eos')

  indx1 = 1
  iic = ntstart
  d12 = 0.0_c_double
  DO istep = 1, nsteps
    s%iic = iic
    s%ntfirst = ntstart
    s%nstp = 1 + MOD(iic-ntstart, 2)          ! main3d.F:189-191
    s%nnew = 3 - s%nstp
    s%nrhs = s%nstp
    IF (iic == ntstart) THEN                  ! main3d.F:269-283
      CALL check (roms_hip_ini_zeta (s), 'ini_zeta')
      CALL check (roms_hip_set_depth (s), 'set_depth')
      CALL check (roms_hip_ini_fields (s), 'ini_fields')
    END IF
    CALL check (roms_hip_set_massflux (s), 'set_massflux')
    CALL check (roms_hip_rho_eos (s), 'rho_eos')
    CALL check (roms_hip_diag (s, d12), 'diag')
    CALL check (roms_hip_omega (s), 'omega')
    CALL check (roms_hip_wvelocity (s), 'wvelocity')
    CALL check (roms_hip_set_zeta (s), 'set_zeta')
    CALL check (roms_hip_rhs3d (s), 'rhs3d')
    CALL check (roms_hip_step2d_loop (s, indx1), 'step2d_loop')
    CALL check (roms_hip_set_depth (s), 'set_depth')
    CALL check (roms_hip_step3d_uv (s), 'step3d_uv')
    CALL check (roms_hip_omega (s), 'omega')
    CALL check (roms_hip_step3d_t (s), 'step3d_t')
    iic = iic + 1
  END DO
  DO q = 1, SIZE(outids)
    CALL check (roms_hip_sync_to_host (outids(q)), 'sync_to_host')
  END DO

  OPEN (11, FILE=TRIM(fout), ACCESS='stream', FORM='unformatted', STATUS='replace')
  WRITE (11) INT(SIZE(outids), c_long), INT(indx1, c_long), INT(s%nnew, c_long)
  DO q = 1, SIZE(outids)
    WRITE (11) INT(outids(q), c_long), INT(SIZE(F(outids(q))%a), c_long)
    WRITE (11) F(outids(q))%a
  END DO
  WRITE (11) d12
  CLOSE (11)
  CALL check (roms_hip_finalize (), 'finalize')
  WRITE (*, '(a,i0,a,1p,e22.15)') 'roms_hip_demo: ', nsteps, ' steps, volume of the last diag = ', d12(1)

CONTAINS

  SUBROUTINE check (rc_, what)
    INTEGER(c_int), INTENT(in) :: rc_
    CHARACTER(len=*), INTENT(in) :: what
    CHARACTER(kind=c_char), POINTER :: msg(:)
    INTEGER :: n
    CALL roms_hip_status (rc_, exit_flag)
    IF (exit_flag /= 0) THEN
      CALL c_f_pointer (roms_hip_last_error (), msg, (/ 400 /))
      n = 1
      DO WHILE (n < 400 .AND. msg(n) /= c_null_char)
        n = n + 1
      END DO
      WRITE (*, *) 'roms_hip_demo: ', what, ' failed, exit_flag = ', exit_flag, ': ', msg(1:n-1)
      STOP 8
    END IF
  END SUBROUTINE check

END PROGRAM roms_hip_demo
