!=======================================================================
!  roms_hip_demo_mpi -- a Fortran + MPI host for libroms_hip.so, used by
!  tests/test_gpu_fortran_host.py: one MPI rank per tile, the halo exchange
!  of the library handed to the host's own MPI (roms_hip_set_halo_relay)
!  exactly where the reference's mp_exchange2d/3d/4d post their
!  MPI_Irecv / MPI_Isend / MPI_Waitall (ROMS/Utility/mp_exchange.F:290-560).
!  It is NOT part of ROMS.
!
!    mpiexec -n NtileI*NtileJ roms_hip_demo_mpi <state_prefix> <result_prefix> <nsteps> <NtileI> <NtileJ>
!=======================================================================
MODULE relay_mod
  USE, INTRINSIC :: iso_c_binding
  USE roms_hip_mod, ONLY : roms_halo_msg_t
  IMPLICIT NONE
  INCLUDE 'mpif.h'
CONTAINS
  FUNCTION mpi_relay (user, nsend, send, nrecv, recv) BIND(C) RESULT(rc)
    TYPE(c_ptr), VALUE :: user
    INTEGER(c_int), VALUE :: nsend, nrecv
    TYPE(roms_halo_msg_t), INTENT(in) :: send(nsend), recv(nrecv)
    INTEGER(c_int) :: rc
    INTEGER :: req(16), m, ierr, nreq
    REAL(c_double), POINTER :: buf(:)
    nreq = 0
    DO m = 1, nrecv
      CALL c_f_pointer (recv(m)%buf, buf, (/ recv(m)%count /))
      nreq = nreq + 1
      CALL MPI_Irecv (buf(1), INT(recv(m)%count), MPI_DOUBLE_PRECISION, recv(m)%peer, recv(m)%tag, MPI_COMM_WORLD, req(nreq), ierr)
    END DO
    DO m = 1, nsend
      CALL c_f_pointer (send(m)%buf, buf, (/ send(m)%count /))
      nreq = nreq + 1
      CALL MPI_Isend (buf(1), INT(send(m)%count), MPI_DOUBLE_PRECISION, send(m)%peer, send(m)%tag, MPI_COMM_WORLD, req(nreq), ierr)
    END DO
    CALL MPI_Waitall (nreq, req, MPI_STATUSES_IGNORE, ierr)
    rc = MERGE(0_c_int, 1_c_int, ierr == MPI_SUCCESS)
  END FUNCTION mpi_relay
END MODULE relay_mod

PROGRAM roms_hip_demo_mpi
  USE, INTRINSIC :: iso_c_binding
  USE roms_hip_mod
  USE relay_mod
  IMPLICIT NONE
  INTEGER, PARAMETER :: MAXF = 256
  TYPE fld
    REAL(c_double), POINTER :: a(:) => NULL()
  END TYPE fld
  TYPE(fld) :: F(0:MAXF-1)
  INTEGER(c_int8_t), ALLOCATABLE, TARGET :: bimg(:), pimg(:)
  INTEGER(c_long) :: nb, np, nf, id, cnt
  INTEGER :: nsteps, istep, q, iic, ntstart, exit_flag, rank, nranks, ierr, ntI, ntJ
  INTEGER(c_int) :: indx1
  TYPE(roms_step_idx_t) :: s
  CHARACTER(len=512) :: pin, pout, arg, fname
  INTEGER(c_int), PARAMETER :: outids(9) = (/ FID_zeta, FID_ubar, FID_vbar, FID_u, FID_v, FID_t, FID_Huon, FID_W, FID_Hz /)

  CALL MPI_Init (ierr)
  CALL MPI_Comm_rank (MPI_COMM_WORLD, rank, ierr)
  CALL MPI_Comm_size (MPI_COMM_WORLD, nranks, ierr)
  CALL get_command_argument (1, pin)
  CALL get_command_argument (2, pout)
  CALL get_command_argument (3, arg); READ (arg, *) nsteps
  CALL get_command_argument (4, arg); READ (arg, *) ntI
  CALL get_command_argument (5, arg); READ (arg, *) ntJ
  IF (ntI*ntJ /= nranks) STOP 3
  exit_flag = 0

  WRITE (fname, '(a,i0,a)') TRIM(pin), rank, '.bin'
  OPEN (10, FILE=TRIM(fname), ACCESS='stream', FORM='unformatted', STATUS='old')
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
  CLOSE (10)

  !  no RCCL id: the halos travel through the relay below (all ranks may share one GPU)
  CALL check (roms_hip_init (INT(rank, c_int), INT(ntI, c_int), INT(ntJ, c_int), 0_c_int, c_null_ptr), 'init')
  CALL check (roms_hip_set_halo_relay (c_funloc(mpi_relay), c_null_ptr), 'set_halo_relay')
  CALL check (roms_hip_set_bounds (c_loc(bimg)), 'set_bounds')
  CALL check (roms_hip_set_params (c_loc(pimg)), 'set_params')
  DO q = 0, MAXF-1
    IF (ASSOCIATED(F(q)%a)) THEN
      CALL check (roms_hip_register_field (INT(q, c_int), c_loc(F(q)%a(1)), INT(SIZE(F(q)%a), c_long)), 'register_field')
    END IF
  END DO
  CALL check (roms_hip_sync_all_to_device (), 'sync_all_to_device')

  ntstart = 1
  s = roms_hip_make_idx (ntstart, ntstart, 1, 2, 1, 1, 1, 1, 1, .FALSE.)
  CALL check (roms_hip_set_depth (s), 'set_depth')
  CALL check (roms_hip_set_massflux (s), 'set_massflux')
  CALL check (roms_hip_omega (s), 'omega')
  CALL check (roms_hip_rho_eos (s), 'rho_eos')
  indx1 = 1
  iic = ntstart
  DO istep = 1, nsteps
    s%iic = iic
    s%ntfirst = ntstart
    s%nstp = 1 + MOD(iic-ntstart, 2)
    s%nnew = 3 - s%nstp
    s%nrhs = s%nstp
    IF (iic == ntstart) THEN                  ! main3d.F:269-283
      CALL check (roms_hip_ini_zeta (s), 'ini_zeta')
      CALL check (roms_hip_set_depth (s), 'set_depth')
      CALL check (roms_hip_ini_fields (s), 'ini_fields')
    END IF
    CALL check (roms_hip_set_massflux (s), 'set_massflux')
    CALL check (roms_hip_rho_eos (s), 'rho_eos')
    CALL check (roms_hip_omega (s), 'omega')
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
  WRITE (fname, '(a,i0,a)') TRIM(pout), rank, '.bin'
  OPEN (11, FILE=TRIM(fname), ACCESS='stream', FORM='unformatted', STATUS='replace')
  WRITE (11) INT(SIZE(outids), c_long), INT(indx1, c_long), INT(s%nnew, c_long)
  DO q = 1, SIZE(outids)
    WRITE (11) INT(outids(q), c_long), INT(SIZE(F(outids(q))%a), c_long)
    WRITE (11) F(outids(q))%a
  END DO
  CLOSE (11)
  CALL check (roms_hip_finalize (), 'finalize')
  CALL MPI_Barrier (MPI_COMM_WORLD, ierr)
  IF (rank == 0) WRITE (*, '(a,i0,a,i0,a)') 'roms_hip_demo_mpi: ', nsteps, ' steps on ', nranks, ' tiles'
  CALL MPI_Finalize (ierr)

CONTAINS

  SUBROUTINE check (rc_, what)
    INTEGER(c_int), INTENT(in) :: rc_
    CHARACTER(len=*), INTENT(in) :: what
    CHARACTER(kind=c_char), POINTER :: msg(:)
    INTEGER :: n, ierr_
    CALL roms_hip_status (rc_, exit_flag)
    IF (exit_flag /= 0) THEN
      CALL c_f_pointer (roms_hip_last_error (), msg, (/ 400 /))
      n = 1
      DO WHILE (n < 400 .AND. msg(n) /= c_null_char)
        n = n + 1
      END DO
      WRITE (*, *) 'roms_hip_demo_mpi: ', what, ' failed on rank ', rank, ': ', msg(1:n-1)
      CALL MPI_Abort (MPI_COMM_WORLD, 8, ierr_)
    END IF
  END SUBROUTINE check

END PROGRAM roms_hip_demo_mpi
