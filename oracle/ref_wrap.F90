!=======================================================================
!  ref_wrap.F90 -- TEST INFRASTRUCTURE ONLY (our code, not the reference's).
!
!  bind(C) doorway to the reference objects that oracle/build_ref.sh compiles
!  from /root/reference with flang.  It drives the reference through ITS OWN
!  public API: allocate_param / initialize_param / allocate_scalars /
!  initialize_scalars / initialize_parallel, get_domain_edges / get_tile /
!  get_bounds exactly as inp_par.F:300-420 does, allocate_grid / _ocean /
!  _coupling / _mixing / _forces, then the module procedures
!  set_depth, set_massflux, set_zeta, rho_eos, prsgrd, t3dmix2, uv3dmix2,
!  set_weights.  Arrays travel in the same roms_fields_t / roms_bounds_t /
!  roms_params_t images as the C oracle uses (include/roms_hip.h).
!  Serial build (no DISTRIBUTE): one tile covering the whole grid.
!=======================================================================
MODULE ref_wrap_types
  USE, INTRINSIC :: iso_c_binding
  IMPLICIT NONE
  TYPE, BIND(C) :: bounds_t
    INTEGER(c_int) :: Lm, Mm, N, NT, NAT
    INTEGER(c_int) :: ntileI, ntileJ, tile, Itile, Jtile
    INTEGER(c_int) :: NghostPoints, EWperiodic, NSperiodic
    INTEGER(c_int) :: west_edge, east_edge, south_edge, north_edge
    INTEGER(c_int) :: LBi, UBi, LBj, UBj
    INTEGER(c_int) :: Istr, Iend, Jstr, Jend
    INTEGER(c_int) :: IstrB, IendB, IstrM, IstrP, IendP, IstrR, IendR, IstrT, IendT, IstrU
    INTEGER(c_int) :: JstrB, JendB, JstrM, JstrP, JendP, JstrR, JendR, JstrT, JendT, JstrV
    INTEGER(c_int) :: Istrm3, Istrm2, Istrm1, IstrUm2, IstrUm1
    INTEGER(c_int) :: Iendp1, Iendp2, Iendp2i, Iendp3
    INTEGER(c_int) :: Jstrm3, Jstrm2, Jstrm1, JstrVm2, JstrVm1
    INTEGER(c_int) :: Jendp1, Jendp2, Jendp2i, Jendp3
  END TYPE bounds_t
  TYPE, BIND(C) :: params_t
    REAL(c_double) :: dt, dtfast, g, rho0, gamma2, lambda
    INTEGER(c_int) :: ndtfast, nfast
    REAL(c_double) :: weight1(256), weight2(256)
    INTEGER(c_int) :: Vtransform, limit_bstress
    REAL(c_double) :: hc
    REAL(c_double) :: sc_r(65), Cs_r(65), sc_w(65), Cs_w(65)
    INTEGER(c_int) :: Hadv(16), Vadv(16)
    INTEGER(c_int) :: lbc_west, lbc_east, lbc_south, lbc_north
    INTEGER(c_int) :: nonlin_eos, eminusp
    REAL(c_double) :: R0, T0, S0, Tcoef, Scoef
    INTEGER(c_int) :: uv_adv, uv_cor, uv_vis2, curvgrid, var_rho_2d
    INTEGER(c_int) :: ts_dif2, mix_geo_ts, mix_s_ts, salinity, lmd_nonlocal, solar_source
    INTEGER(c_int) :: splines_vdiff, splines_vvisc
    REAL(c_double) :: Akt_bak(16), Akv_bak
    REAL(c_double) :: swfrac_mu1, swfrac_mu2, swfrac_r1
    INTEGER(c_int) :: uv_drag, mpdata_fast
    REAL(c_double) :: blk_ZQ, blk_ZT, blk_ZW
    INTEGER(c_int) :: masking, pgf
    INTEGER(c_int) :: lbc(6,4)          ! C: lbc[side][variable]
    REAL(c_double) :: obc_out(6,4), obc_in(6,4)   ! nudging coefficients of RadNud edges (1/s)
    INTEGER(c_int) :: ts_dif4, uv_vis4
    INTEGER(c_int) :: mix_iso_ts, radiation_2d
    REAL(c_double) :: Cdb_min, Cdb_max
    INTEGER(c_int) :: gls_mixing, gls_stability, gls_n2s2_horavg, gls_ri_splines
    REAL(c_double) :: gls_p, gls_m, gls_n, gls_cmu0, gls_c1, gls_c2, gls_c3m, gls_c3p, gls_sigk, gls_sigp, gls_Kmin, gls_Pmin
    REAL(c_double) :: Akk_bak, Akp_bak, Zos
    INTEGER(c_int) :: wet_dry, point_sources
    REAL(c_double) :: Dcrit
    INTEGER(c_int) :: atm_press, press_compensate, ts_mix_stability, ts_mix_min_strat
  END TYPE params_t
  TYPE, BIND(C) :: stepidx_t
    INTEGER(c_int) :: iic, ntfirst, nstp, nnew, nrhs, kstp, krhs, knew, iif, predictor
  END TYPE stepidx_t
  !  order of include/roms_fields.def
  TYPE, BIND(C) :: fields_t
    TYPE(c_ptr) :: zeta, ubar, vbar, rzeta, rubar, rvbar, u, v, t, ru, rv, W, rho, pden
    TYPE(c_ptr) :: h, f, fomn, pm, pn, om_r, on_r, om_u, on_u, om_v, on_v, om_p, on_p, omn
    TYPE(c_ptr) :: pmon_r, pnom_r, pmon_p, pnom_p, pmon_u, pnom_u, pmon_v, pnom_v, dmde, dndx
    TYPE(c_ptr) :: Hz, Huon, Hvom, z_r, z_w
    TYPE(c_ptr) :: DU_avg1, DU_avg2, DV_avg1, DV_avg2, Zt_avg1, rufrc, rvfrc, rhoA, rhoS
    TYPE(c_ptr) :: Akv, Akt, ghats, bvf, alpha, beta, visc2_p, visc2_r, diff2
    TYPE(c_ptr) :: sustr, svstr, bustr, bvstr, srflx, stflx, btflx
    TYPE(c_ptr) :: rdrag2, stflux, btflux, Uwind, Vwind, Tair, Pair, Hair, rain, cloud
    TYPE(c_ptr) :: lrflx, lhflx, shflx, evap, hsbl, rdrag
    TYPE(c_ptr) :: wvel, lonr, latr
    TYPE(c_ptr) :: rmask, umask, vmask, pmask
    TYPE(c_ptr) :: zeta_bry, ubar_bry, vbar_bry, u_bry, v_bry, t_bry
    TYPE(c_ptr) :: visc4_p, visc4_r, diff4
    TYPE(c_ptr) :: ZoBot
    TYPE(c_ptr) :: tke, gls, Lscale, Akk, Akp
    TYPE(c_ptr) :: pmask_wet, rmask_wet, umask_wet, vmask_wet, rmask_wet_avg, pmask_full, rmask_full, umask_full, vmask_full
  END TYPE fields_t
  LOGICAL, SAVE :: have_boundary = .FALSE.      ! allocate_boundary is done once per process
END MODULE ref_wrap_types

!-----------------------------------------------------------------------
FUNCTION ref_abi_sizeof (which) BIND(C, name='ref_abi_sizeof') RESULT(n)
  USE ref_wrap_types
  INTEGER(c_int), VALUE :: which
  INTEGER(c_int) :: n
  TYPE(bounds_t) :: b
  TYPE(params_t) :: p
  TYPE(stepidx_t) :: s
  TYPE(fields_t) :: f
  n = -1
  IF (which == 0) n = INT(c_sizeof(b), c_int)
  IF (which == 1) n = INT(c_sizeof(p), c_int)
  IF (which == 2) n = INT(c_sizeof(s), c_int)
  IF (which == 3) n = INT(c_sizeof(f), c_int)
END FUNCTION ref_abi_sizeof

!-----------------------------------------------------------------------
!  Mirror of the start-up sequence of inp_par.F / read_phypar.F for one grid.
FUNCTION ref_setup (b, p) BIND(C, name='ref_setup') RESULT(rc)
  USE ref_wrap_types
  USE mod_param
  USE mod_parallel
  USE mod_scalars
  USE mod_iounits
  USE mod_stepping, ONLY : allocate_stepping
  USE mod_grid,     ONLY : allocate_grid
  USE mod_ocean,    ONLY : allocate_ocean
  USE mod_coupling, ONLY : allocate_coupling
  USE mod_mixing,   ONLY : allocate_mixing
  USE mod_forces,   ONLY : allocate_forces
  TYPE(bounds_t), INTENT(in) :: b
  TYPE(params_t), INTENT(in) :: p
  INTEGER(c_int) :: rc
  INTEGER :: ng, tile, Itile, Jtile, Nghost, k, LBi, UBi, LBj, UBj
  LOGICAL, SAVE :: done = .FALSE.
  rc = 0
  IF (done) RETURN
  done = .TRUE.
  ng = 1
  Ngrids = 1
  CALL allocate_param
  Lm(ng) = b%Lm;  Mm(ng) = b%Mm;  N(ng) = b%N
  NtileI(ng) = 1; NtileJ(ng) = 1
  NAT = b%NAT
  CALL initialize_param
  CALL initialize_parallel
  CALL allocate_parallel (Ngrids)
  CALL allocate_scalars
  CALL initialize_scalars
  CALL allocate_stepping (Ngrids)
  EWperiodic(ng) = b%EWperiodic /= 0
  NSperiodic(ng) = b%NSperiodic /= 0
  NghostPoints = b%NghostPoints
  !  inp_par.F:340-420
  DO tile = -1, NtileI(ng)*NtileJ(ng)-1
    CALL get_domain_edges (ng, tile,                                         &
 &        DOMAIN(ng) % Eastern_Edge    (tile), DOMAIN(ng) % Western_Edge    (tile), &
 &        DOMAIN(ng) % Northern_Edge   (tile), DOMAIN(ng) % Southern_Edge   (tile), &
 &        DOMAIN(ng) % NorthEast_Corner(tile), DOMAIN(ng) % NorthWest_Corner(tile), &
 &        DOMAIN(ng) % SouthEast_Corner(tile), DOMAIN(ng) % SouthWest_Corner(tile), &
 &        DOMAIN(ng) % NorthEast_Test  (tile), DOMAIN(ng) % NorthWest_Test  (tile), &
 &        DOMAIN(ng) % SouthEast_Test  (tile), DOMAIN(ng) % SouthWest_Test  (tile))
  END DO
  Nghost = NghostPoints
  BOUNDS(ng) % LBij = 0
  BOUNDS(ng) % UBij = MAX(Lm(ng)+1, Mm(ng)+1)
  DO tile = -1, NtileI(ng)*NtileJ(ng)-1
    BOUNDS(ng) % tile(tile) = tile
    CALL get_tile (ng, tile, Itile, Jtile,                                    &
 &     BOUNDS(ng) % Istr(tile),  BOUNDS(ng) % Iend(tile),  BOUNDS(ng) % Jstr(tile),  BOUNDS(ng) % Jend(tile),  &
 &     BOUNDS(ng) % IstrM(tile), BOUNDS(ng) % IstrR(tile), BOUNDS(ng) % IstrU(tile), BOUNDS(ng) % IendR(tile), &
 &     BOUNDS(ng) % JstrM(tile), BOUNDS(ng) % JstrR(tile), BOUNDS(ng) % JstrV(tile), BOUNDS(ng) % JendR(tile), &
 &     BOUNDS(ng) % IstrB(tile), BOUNDS(ng) % IendB(tile), BOUNDS(ng) % IstrP(tile), BOUNDS(ng) % IendP(tile), &
 &     BOUNDS(ng) % IstrT(tile), BOUNDS(ng) % IendT(tile),                                                     &
 &     BOUNDS(ng) % JstrB(tile), BOUNDS(ng) % JendB(tile), BOUNDS(ng) % JstrP(tile), BOUNDS(ng) % JendP(tile), &
 &     BOUNDS(ng) % JstrT(tile), BOUNDS(ng) % JendT(tile),                                                     &
 &     BOUNDS(ng) % Istrm3(tile), BOUNDS(ng) % Istrm2(tile), BOUNDS(ng) % Istrm1(tile),                        &
 &     BOUNDS(ng) % IstrUm2(tile), BOUNDS(ng) % IstrUm1(tile),                                                 &
 &     BOUNDS(ng) % Iendp1(tile), BOUNDS(ng) % Iendp2(tile), BOUNDS(ng) % Iendp2i(tile), BOUNDS(ng) % Iendp3(tile), &
 &     BOUNDS(ng) % Jstrm3(tile), BOUNDS(ng) % Jstrm2(tile), BOUNDS(ng) % Jstrm1(tile),                        &
 &     BOUNDS(ng) % JstrVm2(tile), BOUNDS(ng) % JstrVm1(tile),                                                 &
 &     BOUNDS(ng) % Jendp1(tile), BOUNDS(ng) % Jendp2(tile), BOUNDS(ng) % Jendp2i(tile), BOUNDS(ng) % Jendp3(tile))
    CALL get_bounds (ng, tile, 0, Nghost, Itile, Jtile,                        &
 &     BOUNDS(ng) % LBi(tile), BOUNDS(ng) % UBi(tile), BOUNDS(ng) % LBj(tile), BOUNDS(ng) % UBj(tile))
  END DO
  !  scalars the kernels read (read_phypar.F would set them from roms_*.in)
  dt(ng) = p%dt
  g = p%g
  rho0 = p%rho0
  Vtransform(ng) = p%Vtransform
  hc(ng) = p%hc
  DO k = 1, N(ng)
    SCALARS(ng) % sc_r(k) = p%sc_r(k+1)
    SCALARS(ng) % Cs_r(k) = p%Cs_r(k+1)
  END DO
  DO k = 0, N(ng)
    SCALARS(ng) % sc_w(k) = p%sc_w(k+1)
    SCALARS(ng) % Cs_w(k) = p%Cs_w(k+1)
  END DO
  R0(ng) = p%R0; T0(ng) = p%T0; S0(ng) = p%S0; Tcoef(ng) = p%Tcoef; Scoef(ng) = p%Scoef
  LBi = BOUNDS(ng)%LBi(0); UBi = BOUNDS(ng)%UBi(0); LBj = BOUNDS(ng)%LBj(0); UBj = BOUNDS(ng)%UBj(0)
  CALL allocate_grid (ng, LBi, UBi, LBj, UBj, BOUNDS(ng)%LBij, BOUNDS(ng)%UBij)
  CALL allocate_ocean (ng, LBi, UBi, LBj, UBj)
  CALL allocate_coupling (ng, LBi, UBi, LBj, UBj)
  CALL allocate_mixing (ng, LBi, UBi, LBj, UBj)
  CALL allocate_forces (ng, LBi, UBi, LBj, UBj)
END FUNCTION ref_setup

!-----------------------------------------------------------------------
!  BOUNDS(ng)%...(tile 0) in the order of roms_bounds_t from Istr on.
SUBROUTINE ref_get_bounds (out) BIND(C, name='ref_get_bounds')
  USE ref_wrap_types
  USE mod_param
  INTEGER(c_int), INTENT(out) :: out(0:49)
  INTEGER :: t
  t = 0
  out(0:3)   = (/ BOUNDS(1)%LBi(t), BOUNDS(1)%UBi(t), BOUNDS(1)%LBj(t), BOUNDS(1)%UBj(t) /)
  out(4:7)   = (/ BOUNDS(1)%Istr(t), BOUNDS(1)%Iend(t), BOUNDS(1)%Jstr(t), BOUNDS(1)%Jend(t) /)
  out(8:17)  = (/ BOUNDS(1)%IstrB(t), BOUNDS(1)%IendB(t), BOUNDS(1)%IstrM(t), BOUNDS(1)%IstrP(t),      &
 &               BOUNDS(1)%IendP(t), BOUNDS(1)%IstrR(t), BOUNDS(1)%IendR(t), BOUNDS(1)%IstrT(t),      &
 &               BOUNDS(1)%IendT(t), BOUNDS(1)%IstrU(t) /)
  out(18:27) = (/ BOUNDS(1)%JstrB(t), BOUNDS(1)%JendB(t), BOUNDS(1)%JstrM(t), BOUNDS(1)%JstrP(t),      &
 &               BOUNDS(1)%JendP(t), BOUNDS(1)%JstrR(t), BOUNDS(1)%JendR(t), BOUNDS(1)%JstrT(t),      &
 &               BOUNDS(1)%JendT(t), BOUNDS(1)%JstrV(t) /)
  out(28:32) = (/ BOUNDS(1)%Istrm3(t), BOUNDS(1)%Istrm2(t), BOUNDS(1)%Istrm1(t), BOUNDS(1)%IstrUm2(t), BOUNDS(1)%IstrUm1(t) /)
  out(33:36) = (/ BOUNDS(1)%Iendp1(t), BOUNDS(1)%Iendp2(t), BOUNDS(1)%Iendp2i(t), BOUNDS(1)%Iendp3(t) /)
  out(37:41) = (/ BOUNDS(1)%Jstrm3(t), BOUNDS(1)%Jstrm2(t), BOUNDS(1)%Jstrm1(t), BOUNDS(1)%JstrVm2(t), BOUNDS(1)%JstrVm1(t) /)
  out(42:45) = (/ BOUNDS(1)%Jendp1(t), BOUNDS(1)%Jendp2(t), BOUNDS(1)%Jendp2i(t), BOUNDS(1)%Jendp3(t) /)
  out(46:49) = (/ MERGE(1,0,DOMAIN(1)%Western_Edge(t)), MERGE(1,0,DOMAIN(1)%Eastern_Edge(t)),         &
 &               MERGE(1,0,DOMAIN(1)%Southern_Edge(t)), MERGE(1,0,DOMAIN(1)%Northern_Edge(t)) /)
END SUBROUTINE ref_get_bounds

!-----------------------------------------------------------------------
!  set_weights (ROMS/Utility/set_weights.F:3)
SUBROUTINE ref_set_weights (ndtfast_in, nfast_out, w1, w2) BIND(C, name='ref_set_weights')
  USE ref_wrap_types
  USE mod_param
  USE mod_scalars
  USE mod_iounits
  INTEGER(c_int), VALUE :: ndtfast_in
  INTEGER(c_int), INTENT(out) :: nfast_out
  REAL(c_double), INTENT(out) :: w1(2*ndtfast_in), w2(2*ndtfast_in)
  INTEGER :: i
  ndtfast(1) = ndtfast_in
  IF (allocated(weight)) deallocate (weight)
  allocate ( weight(2, 0:256, 1) )
  weight = 0.0_r8
  LwrtInfo(1) = .FALSE.
  CALL set_weights (1)
  nfast_out = nfast(1)
  DO i = 1, 2*ndtfast_in
    w1(i) = weight(1,i,1)
    w2(i) = weight(2,i,1)
  END DO
END SUBROUTINE ref_set_weights

!-----------------------------------------------------------------------
!  Run one reference module procedure on the arrays of F (copied into the
!  reference's module state, results copied back).
!  kernel: 1 set_depth  2 set_massflux  3 set_zeta  4 rho_eos  5 prsgrd
!          6 t3dmix2    7 uv3dmix2
FUNCTION ref_call (kernel, b, p, s, F) BIND(C, name='ref_call') RESULT(rc)
  USE ref_wrap_types
  USE mod_param
  USE mod_scalars
  USE mod_stepping
  USE mod_grid
  USE mod_ocean
  USE mod_coupling
  USE mod_mixing
  USE set_depth_mod,    ONLY : set_depth
  USE set_massflux_mod, ONLY : set_massflux
  USE set_zeta_mod,     ONLY : set_zeta
  USE rho_eos_mod,      ONLY : rho_eos
  USE prsgrd_mod,       ONLY : prsgrd
#ifdef ATM_PRESS
  USE mod_forces
#endif
  USE t3dmix2_mod,      ONLY : t3dmix2
  USE uv3dmix2_mod,     ONLY : uv3dmix2
#ifdef REF_DIF4
  USE t3dmix4_mod,      ONLY : t3dmix4
  USE uv3dmix4_mod,     ONLY : uv3dmix4
  USE mod_ncparam
#endif
  INTEGER(c_int), VALUE :: kernel
  TYPE(bounds_t), INTENT(in) :: b
  TYPE(params_t), INTENT(in) :: p
  TYPE(stepidx_t), INTENT(in) :: s
  TYPE(fields_t), INTENT(in) :: F
  INTEGER(c_int) :: rc
  INTEGER :: ng, tile, LBi, UBi, LBj, UBj, ni, nj, NN, NTT
  INTEGER :: side4(4), sd4, v4, code4, it4
  REAL(c_double), POINTER :: a2(:,:), a3(:,:,:), a4(:,:,:,:), a5(:,:,:,:,:)
  ng = 1; tile = 0
  LBi = b%LBi; UBi = b%UBi; LBj = b%LBj; UBj = b%UBj
  ni = UBi-LBi+1; nj = UBj-LBj+1; NN = b%N; NTT = b%NT
  rc = 0
  IF (LBi /= BOUNDS(ng)%LBi(0) .OR. UBi /= BOUNDS(ng)%UBi(0) .OR. LBj /= BOUNDS(ng)%LBj(0) .OR. UBj /= BOUNDS(ng)%UBj(0)) THEN
    rc = 1
    RETURN
  END IF
  nstp(ng) = s%nstp; nnew(ng) = s%nnew; nrhs(ng) = s%nrhs
  dt(ng) = p%dt
  ! ---- copy in ----
  CALL c_f_pointer (F%h, a2, (/ni,nj/));        GRID(ng)%h = a2
  CALL c_f_pointer (F%pm, a2, (/ni,nj/));       GRID(ng)%pm = a2
  CALL c_f_pointer (F%pn, a2, (/ni,nj/));       GRID(ng)%pn = a2
  CALL c_f_pointer (F%om_u, a2, (/ni,nj/));     GRID(ng)%om_u = a2
  CALL c_f_pointer (F%on_u, a2, (/ni,nj/));     GRID(ng)%on_u = a2
  CALL c_f_pointer (F%om_v, a2, (/ni,nj/));     GRID(ng)%om_v = a2
  CALL c_f_pointer (F%on_v, a2, (/ni,nj/));     GRID(ng)%on_v = a2
  CALL c_f_pointer (F%om_r, a2, (/ni,nj/));     GRID(ng)%om_r = a2
  CALL c_f_pointer (F%on_r, a2, (/ni,nj/));     GRID(ng)%on_r = a2
  CALL c_f_pointer (F%om_p, a2, (/ni,nj/));     GRID(ng)%om_p = a2
  CALL c_f_pointer (F%on_p, a2, (/ni,nj/));     GRID(ng)%on_p = a2
  CALL c_f_pointer (F%pmon_r, a2, (/ni,nj/));   GRID(ng)%pmon_r = a2
  CALL c_f_pointer (F%pnom_r, a2, (/ni,nj/));   GRID(ng)%pnom_r = a2
  CALL c_f_pointer (F%pmon_p, a2, (/ni,nj/));   GRID(ng)%pmon_p = a2
  CALL c_f_pointer (F%pnom_p, a2, (/ni,nj/));   GRID(ng)%pnom_p = a2
  CALL c_f_pointer (F%pmon_u, a2, (/ni,nj/));   GRID(ng)%pmon_u = a2
  CALL c_f_pointer (F%pnom_v, a2, (/ni,nj/));   GRID(ng)%pnom_v = a2
#ifdef ATM_PRESS
  CALL c_f_pointer (F%Pair, a2, (/ni,nj/));     FORCES(ng)%Pair = a2
#endif
#ifdef MASKING
  CALL c_f_pointer (F%rmask, a2, (/ni,nj/));    GRID(ng)%rmask = a2
  CALL c_f_pointer (F%umask, a2, (/ni,nj/));    GRID(ng)%umask = a2
  CALL c_f_pointer (F%vmask, a2, (/ni,nj/));    GRID(ng)%vmask = a2
  CALL c_f_pointer (F%pmask, a2, (/ni,nj/));    GRID(ng)%pmask = a2
#endif
#ifdef WET_DRY
  CALL c_f_pointer (F%rmask_wet, a2, (/ni,nj/)); GRID(ng)%rmask_wet = a2
  CALL c_f_pointer (F%umask_wet, a2, (/ni,nj/)); GRID(ng)%umask_wet = a2
  CALL c_f_pointer (F%vmask_wet, a2, (/ni,nj/)); GRID(ng)%vmask_wet = a2
  CALL c_f_pointer (F%pmask_wet, a2, (/ni,nj/)); GRID(ng)%pmask_wet = a2
  Dcrit(ng) = p%Dcrit
#endif
  CALL c_f_pointer (F%Hz, a3, (/ni,nj,NN/));    GRID(ng)%Hz = a3
  CALL c_f_pointer (F%Huon, a3, (/ni,nj,NN/));  GRID(ng)%Huon = a3
  CALL c_f_pointer (F%Hvom, a3, (/ni,nj,NN/));  GRID(ng)%Hvom = a3
  CALL c_f_pointer (F%z_r, a3, (/ni,nj,NN/));   GRID(ng)%z_r = a3
  CALL c_f_pointer (F%z_w, a3, (/ni,nj,NN+1/)); GRID(ng)%z_w = a3
  CALL c_f_pointer (F%zeta, a3, (/ni,nj,3/));   OCEAN(ng)%zeta = a3
  CALL c_f_pointer (F%u, a4, (/ni,nj,NN,2/));   OCEAN(ng)%u = a4
  CALL c_f_pointer (F%v, a4, (/ni,nj,NN,2/));   OCEAN(ng)%v = a4
  CALL c_f_pointer (F%ru, a4, (/ni,nj,NN+1,2/)); OCEAN(ng)%ru = a4
  CALL c_f_pointer (F%rv, a4, (/ni,nj,NN+1,2/)); OCEAN(ng)%rv = a4
  CALL c_f_pointer (F%t, a5, (/ni,nj,NN,3,NTT/)); OCEAN(ng)%t = a5
  CALL c_f_pointer (F%rho, a3, (/ni,nj,NN/));   OCEAN(ng)%rho = a3
  CALL c_f_pointer (F%pden, a3, (/ni,nj,NN/));  OCEAN(ng)%pden = a3
  CALL c_f_pointer (F%Zt_avg1, a2, (/ni,nj/));  COUPLING(ng)%Zt_avg1 = a2
  CALL c_f_pointer (F%rufrc, a2, (/ni,nj/));    COUPLING(ng)%rufrc = a2
  CALL c_f_pointer (F%rvfrc, a2, (/ni,nj/));    COUPLING(ng)%rvfrc = a2
  CALL c_f_pointer (F%rhoA, a2, (/ni,nj/));     COUPLING(ng)%rhoA = a2
  CALL c_f_pointer (F%rhoS, a2, (/ni,nj/));     COUPLING(ng)%rhoS = a2
#if defined BENCHMARK || defined UPWELLING || defined REF_GEOUV
  CALL c_f_pointer (F%visc2_p, a2, (/ni,nj/));  MIXING(ng)%visc2_p = a2
  CALL c_f_pointer (F%visc2_r, a2, (/ni,nj/));  MIXING(ng)%visc2_r = a2
#endif
  CALL c_f_pointer (F%diff2, a3, (/ni,nj,NTT/)); MIXING(ng)%diff2 = a3
#ifdef REF_DIF4
  CALL c_f_pointer (F%visc4_p, a2, (/ni,nj/));  MIXING(ng)%visc4_p = a2
  CALL c_f_pointer (F%visc4_r, a2, (/ni,nj/));  MIXING(ng)%visc4_r = a2
  CALL c_f_pointer (F%diff4, a3, (/ni,nj,NTT/)); MIXING(ng)%diff4 = a3
  gamma2(ng) = p%gamma2
  !  the biharmonic operators ask whether the condition of the variable on a physical edge is "closed"
  !  (t3dmix4_s.h:353, uv3dmix4_s.h:392): LBC from p%lbc (rows 4, 5, 6 = u, v, tracers; 0 = the side's lbc_west ...)
  IF (.NOT. allocated(isTvar)) THEN
    allocate ( isTvar(MT) )
    DO it4 = 1, MT
      isTvar(it4) = 5 + it4                     ! mod_ncparam.F:1196-1203
    END DO
  END IF
  side4(1) = iwest; side4(2) = ieast; side4(3) = isouth; side4(4) = inorth
  DO sd4 = 1, 4
    DO v4 = 4, 6
      code4 = p%lbc(v4, sd4)
      IF (code4 == 0) THEN
        IF (sd4 == 1) code4 = p%lbc_west
        IF (sd4 == 2) code4 = p%lbc_east
        IF (sd4 == 3) code4 = p%lbc_south
        IF (sd4 == 4) code4 = p%lbc_north
      END IF
      IF (v4 < 6) THEN
        LBC(side4(sd4), v4, ng)%closed = code4 == 1
      ELSE
        DO it4 = 1, NTT
          LBC(side4(sd4), isTvar(it4), ng)%closed = code4 == 1
        END DO
      END IF
    END DO
  END DO
#endif
#ifdef BENCHMARK
  CALL c_f_pointer (F%bvf, a3, (/ni,nj,NN+1/)); MIXING(ng)%bvf = a3
  CALL c_f_pointer (F%alpha, a2, (/ni,nj/));    MIXING(ng)%alpha = a2
  CALL c_f_pointer (F%beta, a2, (/ni,nj/));     MIXING(ng)%beta = a2
#endif
  ! ---- the reference procedure ----
  SELECT CASE (kernel)
  CASE (1); CALL set_depth (ng, tile, iNLM)
  CASE (2); CALL set_massflux (ng, tile, iNLM)
  CASE (3); CALL set_zeta (ng, tile)
  CASE (4); CALL rho_eos (ng, tile, iNLM)
  CASE (5); CALL prsgrd (ng, tile)
  CASE (6); CALL t3dmix2 (ng, tile)
#if defined BENCHMARK || defined UPWELLING || defined REF_GEOUV
  CASE (7); CALL uv3dmix2 (ng, tile)
#endif
#ifdef REF_DIF4
  CASE (8); CALL t3dmix4 (ng, tile)
  CASE (9); CALL uv3dmix4 (ng, tile)
#endif
  CASE DEFAULT; rc = 2
  END SELECT
  ! ---- copy out ----
  CALL c_f_pointer (F%h, a2, (/ni,nj/));        a2 = GRID(ng)%h
  CALL c_f_pointer (F%Hz, a3, (/ni,nj,NN/));    a3 = GRID(ng)%Hz
  CALL c_f_pointer (F%Huon, a3, (/ni,nj,NN/));  a3 = GRID(ng)%Huon
  CALL c_f_pointer (F%Hvom, a3, (/ni,nj,NN/));  a3 = GRID(ng)%Hvom
  CALL c_f_pointer (F%z_r, a3, (/ni,nj,NN/));   a3 = GRID(ng)%z_r
  CALL c_f_pointer (F%z_w, a3, (/ni,nj,NN+1/)); a3 = GRID(ng)%z_w
  CALL c_f_pointer (F%zeta, a3, (/ni,nj,3/));   a3 = OCEAN(ng)%zeta
  CALL c_f_pointer (F%u, a4, (/ni,nj,NN,2/));   a4 = OCEAN(ng)%u
  CALL c_f_pointer (F%v, a4, (/ni,nj,NN,2/));   a4 = OCEAN(ng)%v
  CALL c_f_pointer (F%ru, a4, (/ni,nj,NN+1,2/)); a4 = OCEAN(ng)%ru
  CALL c_f_pointer (F%rv, a4, (/ni,nj,NN+1,2/)); a4 = OCEAN(ng)%rv
  CALL c_f_pointer (F%t, a5, (/ni,nj,NN,3,NTT/)); a5 = OCEAN(ng)%t
  CALL c_f_pointer (F%rho, a3, (/ni,nj,NN/));   a3 = OCEAN(ng)%rho
  CALL c_f_pointer (F%pden, a3, (/ni,nj,NN/));  a3 = OCEAN(ng)%pden
  CALL c_f_pointer (F%rufrc, a2, (/ni,nj/));    a2 = COUPLING(ng)%rufrc
  CALL c_f_pointer (F%rvfrc, a2, (/ni,nj/));    a2 = COUPLING(ng)%rvfrc
  CALL c_f_pointer (F%rhoA, a2, (/ni,nj/));     a2 = COUPLING(ng)%rhoA
  CALL c_f_pointer (F%rhoS, a2, (/ni,nj/));     a2 = COUPLING(ng)%rhoS
#ifdef BENCHMARK
  CALL c_f_pointer (F%bvf, a3, (/ni,nj,NN+1/)); a3 = MIXING(ng)%bvf
  CALL c_f_pointer (F%alpha, a2, (/ni,nj/));    a2 = MIXING(ng)%alpha
  CALL c_f_pointer (F%beta, a2, (/ni,nj/));     a2 = MIXING(ng)%beta
#endif
END FUNCTION ref_call

!-----------------------------------------------------------------------
!  Per-step physics between the hot kernels: kernel 1 = set_vbc (set_vbc.F:34),
!  2 = bulk_flux (bulk_flux.F:46, BULK_FLUXES applications only = BENCHMARK here).  This file is
!  preprocessed with -D<APP> only, so the application's options are spelled out by application:
!  UV_QDRAG: BENCHMARK, SEAMOUNT; UV_LDRAG: UPWELLING; BULK_FLUXES: BENCHMARK.
FUNCTION ref_physics (kernel, b, p, s, F) BIND(C, name='ref_physics') RESULT(rc)
  USE ref_wrap_types
  USE mod_param
  USE mod_scalars
  USE mod_ncparam
  USE mod_stepping
  USE mod_grid
  USE mod_ocean
  USE mod_mixing
  USE mod_forces
  USE mod_boundary, ONLY : allocate_boundary
  USE set_vbc_mod,   ONLY : set_vbc
#ifdef BENCHMARK
  USE bulk_flux_mod, ONLY : bulk_flux
# ifndef REF_GLS
  USE lmd_vmix_mod,  ONLY : lmd_vmix
# endif
#endif
  INTEGER(c_int), VALUE :: kernel
  TYPE(bounds_t), INTENT(in) :: b
  TYPE(params_t), INTENT(in) :: p
  TYPE(stepidx_t), INTENT(in) :: s
  TYPE(fields_t), INTENT(in) :: F
  INTEGER(c_int) :: rc
  INTEGER :: ng, tile, LBi, UBi, LBj, UBj, ni, nj, NN, NTT
  REAL(c_double), POINTER :: a2(:,:), a3(:,:,:), a4(:,:,:,:), a5(:,:,:,:,:)
  ng = 1; tile = 0
  LBi = b%LBi; UBi = b%UBi; LBj = b%LBj; UBj = b%UBj
  ni = UBi-LBi+1; nj = UBj-LBj+1; NN = b%N; NTT = b%NT
  rc = 0
  IF (.NOT. have_boundary) THEN
    CALL allocate_boundary (ng)          ! LBC_apply switches read by bc_u2d_tile / bc_v2d_tile
    have_boundary = .TRUE.
  END IF
  nstp(ng) = s%nstp; nnew(ng) = s%nnew; nrhs(ng) = s%nrhs
  rho0 = p%rho0; g = p%g
  gamma2(ng) = p%gamma2
  !  generic LBC indices (mod_ncparam.F:1229-1232) and the closed S/N walls of the supported set-up
  isBu2d = isUbar; isBv2d = isVbar
  isBr2d = isFsur; isBw3d = 6                      ! 6 = isTvar(1) (mod_ncparam.F:1196-1203)
  LBC(isouth, isBr2d, ng)%closed = p%lbc_south == 1
  LBC(inorth, isBr2d, ng)%closed = p%lbc_north == 1
  LBC(isouth, isBw3d, ng)%closed = p%lbc_south == 1
  LBC(inorth, isBw3d, ng)%closed = p%lbc_north == 1
  LBC(iwest, isBw3d, ng)%closed = p%lbc_west == 1;  LBC(ieast, isBw3d, ng)%closed = p%lbc_east == 1
  LBC(isouth, isBu2d, ng)%closed = p%lbc_south == 1
  LBC(inorth, isBu2d, ng)%closed = p%lbc_north == 1
  LBC(isouth, isBv2d, ng)%closed = p%lbc_south == 1
  LBC(inorth, isBv2d, ng)%closed = p%lbc_north == 1
  LBC(iwest, isBr2d, ng)%closed = p%lbc_west == 1;  LBC(ieast, isBr2d, ng)%closed = p%lbc_east == 1
  LBC(iwest, isBw3d, ng)%closed = p%lbc_west == 1;  LBC(ieast, isBw3d, ng)%closed = p%lbc_east == 1
  LBC(iwest, isBu2d, ng)%closed = p%lbc_west == 1;  LBC(ieast, isBu2d, ng)%closed = p%lbc_east == 1
  LBC(iwest, isBv2d, ng)%closed = p%lbc_west == 1;  LBC(ieast, isBv2d, ng)%closed = p%lbc_east == 1
  CALL c_f_pointer (F%Hz, a3, (/ni,nj,NN/));    GRID(ng)%Hz = a3
  CALL c_f_pointer (F%z_r, a3, (/ni,nj,NN/));   GRID(ng)%z_r = a3
  CALL c_f_pointer (F%z_w, a3, (/ni,nj,NN+1/)); GRID(ng)%z_w = a3
  CALL c_f_pointer (F%u, a4, (/ni,nj,NN,2/));   OCEAN(ng)%u = a4
  CALL c_f_pointer (F%v, a4, (/ni,nj,NN,2/));   OCEAN(ng)%v = a4
  CALL c_f_pointer (F%t, a5, (/ni,nj,NN,3,NTT/)); OCEAN(ng)%t = a5
  CALL c_f_pointer (F%rho, a3, (/ni,nj,NN/));   OCEAN(ng)%rho = a3
#ifdef MASKING
  CALL c_f_pointer (F%rmask, a2, (/ni,nj/));    GRID(ng)%rmask = a2
  CALL c_f_pointer (F%umask, a2, (/ni,nj/));    GRID(ng)%umask = a2
  CALL c_f_pointer (F%vmask, a2, (/ni,nj/));    GRID(ng)%vmask = a2
#endif
#ifdef WET_DRY
  CALL c_f_pointer (F%rmask_wet, a2, (/ni,nj/)); GRID(ng)%rmask_wet = a2
  CALL c_f_pointer (F%umask_wet, a2, (/ni,nj/)); GRID(ng)%umask_wet = a2
  CALL c_f_pointer (F%vmask_wet, a2, (/ni,nj/)); GRID(ng)%vmask_wet = a2
  CALL c_f_pointer (F%pmask_wet, a2, (/ni,nj/)); GRID(ng)%pmask_wet = a2
  Dcrit(ng) = p%Dcrit
#endif
#if defined BENCHMARK || defined SEAMOUNT
  CALL c_f_pointer (F%rdrag2, a2, (/ni,nj/));   GRID(ng)%rdrag2 = a2
#endif
#if defined UPWELLING && !defined REF_LOGDRAG
  CALL c_f_pointer (F%rdrag, a2, (/ni,nj/));    GRID(ng)%rdrag = a2
#endif
#ifdef REF_LOGDRAG
  CALL c_f_pointer (F%ZoBot, a2, (/ni,nj/));    GRID(ng)%ZoBot = a2
  Cdb_min = p%Cdb_min; Cdb_max = p%Cdb_max
#endif
  CALL c_f_pointer (F%bustr, a2, (/ni,nj/));    FORCES(ng)%bustr = a2
  CALL c_f_pointer (F%bvstr, a2, (/ni,nj/));    FORCES(ng)%bvstr = a2
  CALL c_f_pointer (F%sustr, a2, (/ni,nj/));    FORCES(ng)%sustr = a2
  CALL c_f_pointer (F%svstr, a2, (/ni,nj/));    FORCES(ng)%svstr = a2
  CALL c_f_pointer (F%stflux, a3, (/ni,nj,NTT/)); FORCES(ng)%stflux = a3
  CALL c_f_pointer (F%btflux, a3, (/ni,nj,NTT/)); FORCES(ng)%btflux = a3
  CALL c_f_pointer (F%stflx, a3, (/ni,nj,NTT/)); FORCES(ng)%stflx = a3
  CALL c_f_pointer (F%btflx, a3, (/ni,nj,NTT/)); FORCES(ng)%btflx = a3
#ifdef BENCHMARK
  blk_ZQ(ng) = p%blk_ZQ; blk_ZT(ng) = p%blk_ZT; blk_ZW(ng) = p%blk_ZW
  CALL c_f_pointer (F%alpha, a2, (/ni,nj/));    MIXING(ng)%alpha = a2
  CALL c_f_pointer (F%beta, a2, (/ni,nj/));     MIXING(ng)%beta = a2
  CALL c_f_pointer (F%srflx, a2, (/ni,nj/));    FORCES(ng)%srflx = a2
  CALL c_f_pointer (F%Uwind, a2, (/ni,nj/));    FORCES(ng)%Uwind = a2
  CALL c_f_pointer (F%Vwind, a2, (/ni,nj/));    FORCES(ng)%Vwind = a2
  CALL c_f_pointer (F%Tair, a2, (/ni,nj/));     FORCES(ng)%Tair = a2
  CALL c_f_pointer (F%Pair, a2, (/ni,nj/));     FORCES(ng)%Pair = a2
  CALL c_f_pointer (F%Hair, a2, (/ni,nj/));     FORCES(ng)%Hair = a2
  CALL c_f_pointer (F%rain, a2, (/ni,nj/));     FORCES(ng)%rain = a2
  CALL c_f_pointer (F%cloud, a2, (/ni,nj/));    FORCES(ng)%cloud = a2
  CALL c_f_pointer (F%lrflx, a2, (/ni,nj/));    FORCES(ng)%lrflx = a2
  CALL c_f_pointer (F%lhflx, a2, (/ni,nj/));    FORCES(ng)%lhflx = a2
  CALL c_f_pointer (F%shflx, a2, (/ni,nj/));    FORCES(ng)%shflx = a2
  !  lmd_vmix (LMD_MIXING + LMD_SKPP): uniform Jerlov water type WTYPE = 1 (roms_benchmark*.in:392)
# ifndef REF_GLS
  MIXING(ng)%Jwtype = 1.0_r8
# endif
  iic(ng) = s%iic; ntstart(ng) = s%ntfirst
  CALL c_f_pointer (F%f, a2, (/ni,nj/));        GRID(ng)%f = a2
  CALL c_f_pointer (F%pden, a3, (/ni,nj,NN/));  OCEAN(ng)%pden = a3
  CALL c_f_pointer (F%bvf, a3, (/ni,nj,NN+1/)); MIXING(ng)%bvf = a3
  CALL c_f_pointer (F%Akv, a3, (/ni,nj,NN+1/)); MIXING(ng)%Akv = a3
  CALL c_f_pointer (F%Akt, a4, (/ni,nj,NN+1,INT(b%NAT)/));   MIXING(ng)%Akt = a4
# ifndef REF_GLS
  CALL c_f_pointer (F%ghats, a4, (/ni,nj,NN+1,INT(b%NAT)/)); MIXING(ng)%ghats = a4
  CALL c_f_pointer (F%hsbl, a2, (/ni,nj/));     MIXING(ng)%hsbl = a2
# endif
#endif
  SELECT CASE (kernel)
  CASE (1); CALL set_vbc (ng, tile)
#ifdef BENCHMARK
  CASE (2); CALL bulk_flux (ng, tile)
# ifndef REF_GLS
  CASE (3); CALL lmd_vmix (ng, tile)
# endif
#endif
  CASE DEFAULT; rc = 2
  END SELECT
  CALL c_f_pointer (F%bustr, a2, (/ni,nj/));    a2 = FORCES(ng)%bustr
  CALL c_f_pointer (F%bvstr, a2, (/ni,nj/));    a2 = FORCES(ng)%bvstr
  CALL c_f_pointer (F%sustr, a2, (/ni,nj/));    a2 = FORCES(ng)%sustr
  CALL c_f_pointer (F%svstr, a2, (/ni,nj/));    a2 = FORCES(ng)%svstr
  CALL c_f_pointer (F%stflux, a3, (/ni,nj,NTT/)); a3 = FORCES(ng)%stflux
  CALL c_f_pointer (F%stflx, a3, (/ni,nj,NTT/)); a3 = FORCES(ng)%stflx
  CALL c_f_pointer (F%btflx, a3, (/ni,nj,NTT/)); a3 = FORCES(ng)%btflx
#ifdef BENCHMARK
  CALL c_f_pointer (F%lrflx, a2, (/ni,nj/));    a2 = FORCES(ng)%lrflx
  CALL c_f_pointer (F%lhflx, a2, (/ni,nj/));    a2 = FORCES(ng)%lhflx
  CALL c_f_pointer (F%shflx, a2, (/ni,nj/));    a2 = FORCES(ng)%shflx
#ifdef EMINUSP
  CALL c_f_pointer (F%evap, a2, (/ni,nj/));     a2 = FORCES(ng)%evap
#endif
  CALL c_f_pointer (F%Akv, a3, (/ni,nj,NN+1/)); a3 = MIXING(ng)%Akv
  CALL c_f_pointer (F%Akt, a4, (/ni,nj,NN+1,INT(b%NAT)/));   a4 = MIXING(ng)%Akt
# ifndef REF_GLS
  CALL c_f_pointer (F%ghats, a4, (/ni,nj,NN+1,INT(b%NAT)/)); a4 = MIXING(ng)%ghats
  CALL c_f_pointer (F%hsbl, a2, (/ni,nj/));     a2 = MIXING(ng)%hsbl
# endif
#endif
END FUNCTION ref_physics

!-----------------------------------------------------------------------
!  The two diagnostics main3d runs every step: kernel 1 = wvelocity (wvelocity.F:27, called as
!  main3d.F:475 does: Ninp = nstp), 2 = diag (diag.F:31).  diag keeps nothing: it prints its results
!  (diag.F:449-475, formats 1pe14.6 / 1pe13.6) and resets the sums, so its report is sent to the file
!  ref_diag_stdout.txt in the working directory -- the reference's own output is the only observable.
!  SEAMOUNT is built from ref_headers/seamount_nodiag.h (ANA_DIAG off: Functionals/ana_diag.h does not compile).
FUNCTION ref_diagnostics (kernel, b, p, s, F) BIND(C, name='ref_diagnostics') RESULT(rc)
  USE ref_wrap_types
  USE mod_param
  USE mod_parallel
  USE mod_iounits
  USE mod_scalars
  USE mod_ncparam
  USE mod_stepping
  USE mod_grid
  USE mod_ocean
  USE mod_coupling
  USE mod_boundary, ONLY : allocate_boundary
  USE wvelocity_mod, ONLY : wvelocity
  USE diag_mod,      ONLY : diag
  INTEGER(c_int), VALUE :: kernel
  TYPE(bounds_t), INTENT(in) :: b
  TYPE(params_t), INTENT(in) :: p
  TYPE(stepidx_t), INTENT(in) :: s
  TYPE(fields_t), INTENT(in) :: F
  INTEGER(c_int) :: rc
  INTEGER :: ng, tile, LBi, UBi, LBj, UBj, ni, nj, NN, saved_stdout
  REAL(c_double), POINTER :: a2(:,:), a3(:,:,:), a4(:,:,:,:)
  ng = 1; tile = 0
  LBi = b%LBi; UBi = b%UBi; LBj = b%LBj; UBj = b%UBj
  ni = UBi-LBi+1; nj = UBj-LBj+1; NN = b%N
  rc = 0
  IF (.NOT. have_boundary) THEN
    CALL allocate_boundary (ng)
    have_boundary = .TRUE.
  END IF
  nstp(ng) = s%nstp; nnew(ng) = s%nnew; nrhs(ng) = s%nrhs; krhs(ng) = s%krhs
  iic(ng) = s%iic; ntstart(ng) = s%ntfirst; ninfo(ng) = 1
  rho0 = p%rho0; g = p%g; dt(ng) = p%dt
  isBw3d = 6
  LBC(isouth, isBw3d, ng)%closed = p%lbc_south == 1
  LBC(inorth, isBw3d, ng)%closed = p%lbc_north == 1
  LBC(iwest, isBw3d, ng)%closed = p%lbc_west == 1;  LBC(ieast, isBw3d, ng)%closed = p%lbc_east == 1
  CALL c_f_pointer (F%h, a2, (/ni,nj/));        GRID(ng)%h = a2
  CALL c_f_pointer (F%pm, a2, (/ni,nj/));       GRID(ng)%pm = a2
  CALL c_f_pointer (F%pn, a2, (/ni,nj/));       GRID(ng)%pn = a2
  CALL c_f_pointer (F%omn, a2, (/ni,nj/));      GRID(ng)%omn = a2
  CALL c_f_pointer (F%Hz, a3, (/ni,nj,NN/));    GRID(ng)%Hz = a3
  CALL c_f_pointer (F%z_r, a3, (/ni,nj,NN/));   GRID(ng)%z_r = a3
  CALL c_f_pointer (F%z_w, a3, (/ni,nj,NN+1/)); GRID(ng)%z_w = a3
  CALL c_f_pointer (F%u, a4, (/ni,nj,NN,2/));   OCEAN(ng)%u = a4
  CALL c_f_pointer (F%v, a4, (/ni,nj,NN,2/));   OCEAN(ng)%v = a4
  CALL c_f_pointer (F%rho, a3, (/ni,nj,NN/));   OCEAN(ng)%rho = a3
  CALL c_f_pointer (F%W, a3, (/ni,nj,NN+1/));   OCEAN(ng)%W = a3
  CALL c_f_pointer (F%wvel, a3, (/ni,nj,NN+1/)); OCEAN(ng)%wvel = a3
  CALL c_f_pointer (F%zeta, a3, (/ni,nj,3/));   OCEAN(ng)%zeta = a3
  CALL c_f_pointer (F%ubar, a3, (/ni,nj,3/));   OCEAN(ng)%ubar = a3
  CALL c_f_pointer (F%vbar, a3, (/ni,nj,3/));   OCEAN(ng)%vbar = a3
  CALL c_f_pointer (F%DU_avg1, a2, (/ni,nj/));  COUPLING(ng)%DU_avg1 = a2
  CALL c_f_pointer (F%DV_avg1, a2, (/ni,nj/));  COUPLING(ng)%DV_avg1 = a2
  SELECT CASE (kernel)
  CASE (1)
    CALL wvelocity (ng, tile, nstp(ng))
  CASE (2)
    saved_stdout = stdout
    stdout = 77
    OPEN (UNIT=77, FILE='ref_diag_stdout.txt', STATUS='REPLACE', ACTION='WRITE')
    time_code(ng) = '0001-01-01 00:00:00.00'
    max_speed = 1.0E+30_dp; max_rho = 1.0E+30_dp
    CALL diag (ng, tile)
    CLOSE (77)
    stdout = saved_stdout
  CASE DEFAULT; rc = 2
  END SELECT
  CALL c_f_pointer (F%wvel, a3, (/ni,nj,NN+1/)); a3 = OCEAN(ng)%wvel
  CALL c_f_pointer (F%DU_avg1, a2, (/ni,nj/));  a2 = COUPLING(ng)%DU_avg1
  CALL c_f_pointer (F%DV_avg1, a2, (/ni,nj/));  a2 = COUPLING(ng)%DV_avg1
END FUNCTION ref_diagnostics

#ifndef MASKING      /* analytical.F does not compile with MASKING (ana_mask.h: "no values provided for mask") */
!-----------------------------------------------------------------------
!  The analytic set-up of the application, through the reference's own Functionals (analytical.F) and
!  Utility routines -- used to pin roms_trunk_mgh_amd/ana.py (the inputs of every test and of bench.py):
!    kernel 1  ana_grid + metrics        -> the 2-D GRID fields (h, f, pm, pn, the metric combinations)
!    kernel 2  set_scoord                -> sc_r, Cs_r, sc_w, Cs_w in scout(4*(N+1))
!    kernel 3  ana_initial               -> zeta, ubar, vbar, u, v, t (needs the GRID fields and z_r, z_w, Hz of F)
!    kernel 5  ana_srflux (BENCHMARK: the ALBEDO branch) at tdays = cfg(5) with TIME_REF = 0 as in roms_benchmark*.in;
!              scout(1:2) = the day of the year and the hour caldate returns for it
!    kernel 4  the ana_* forcing of the application (BENCHMARK: winds, tair, pair, humid, rain, cloud;
!              UPWELLING: smflux, stflux(itemp))
!  cfg = theta_s, theta_b, Tcline, Vstretching, tdays.
FUNCTION ref_ana (kernel, b, p, F, cfg, scout) BIND(C, name='ref_ana') RESULT(rc)
  USE ref_wrap_types
  USE mod_param
  USE mod_parallel
  USE mod_iounits
  USE mod_scalars
  USE mod_ncparam
  USE mod_stepping
  USE mod_grid
  USE mod_ocean
  USE mod_forces
  USE mod_mixing
  USE analytical_mod
  USE metrics_mod, ONLY : metrics
  USE dateclock_mod, ONLY : caldate, ref_clock
  INTEGER(c_int), VALUE :: kernel
  TYPE(bounds_t), INTENT(in) :: b
  TYPE(params_t), INTENT(in) :: p
  TYPE(fields_t), INTENT(in) :: F
  REAL(c_double), INTENT(in) :: cfg(5)
  REAL(c_double), INTENT(out) :: scout(*)
  INTEGER(c_int) :: rc
  INTEGER :: ng, tile, LBi, UBi, LBj, UBj, ni, nj, NN, NTT, k, saved_stdout
  REAL(c_double), POINTER :: a2(:,:), a3(:,:,:), a4(:,:,:,:), a5(:,:,:,:,:)
  ng = 1; tile = 0
  LBi = b%LBi; UBi = b%UBi; LBj = b%LBj; UBj = b%UBj
  ni = UBi-LBi+1; nj = UBj-LBj+1; NN = b%N; NTT = b%NT
  rc = 0
  theta_s(ng) = cfg(1); theta_b(ng) = cfg(2); Tcline(ng) = cfg(3); Vstretching(ng) = INT(cfg(4))
  Vtransform(ng) = p%Vtransform
  tdays(ng) = cfg(5); time(ng) = cfg(5)*86400.0_dp
  dt(ng) = p%dt; ndtfast(ng) = p%ndtfast; g = p%g; rho0 = p%rho0
  nstp(ng) = 1; nnew(ng) = 1; nrhs(ng) = 1; kstp(ng) = 1; krhs(ng) = 1; knew(ng) = 1
  saved_stdout = stdout
  stdout = 78                                   ! the routines report statistics: sent to a scratch file
  OPEN (UNIT=78, STATUS='SCRATCH')
  SELECT CASE (kernel)
  CASE (1)
    CALL ana_grid (ng, tile, iNLM)
    CALL metrics (ng, tile, iNLM)
    CALL c_f_pointer (F%h, a2, (/ni,nj/));       a2 = GRID(ng)%h
    CALL c_f_pointer (F%f, a2, (/ni,nj/));       a2 = GRID(ng)%f
    CALL c_f_pointer (F%fomn, a2, (/ni,nj/));    a2 = GRID(ng)%fomn
    CALL c_f_pointer (F%pm, a2, (/ni,nj/));      a2 = GRID(ng)%pm
    CALL c_f_pointer (F%pn, a2, (/ni,nj/));      a2 = GRID(ng)%pn
    CALL c_f_pointer (F%om_r, a2, (/ni,nj/));    a2 = GRID(ng)%om_r
    CALL c_f_pointer (F%on_r, a2, (/ni,nj/));    a2 = GRID(ng)%on_r
    CALL c_f_pointer (F%om_u, a2, (/ni,nj/));    a2 = GRID(ng)%om_u
    CALL c_f_pointer (F%on_u, a2, (/ni,nj/));    a2 = GRID(ng)%on_u
    CALL c_f_pointer (F%om_v, a2, (/ni,nj/));    a2 = GRID(ng)%om_v
    CALL c_f_pointer (F%on_v, a2, (/ni,nj/));    a2 = GRID(ng)%on_v
    CALL c_f_pointer (F%om_p, a2, (/ni,nj/));    a2 = GRID(ng)%om_p
    CALL c_f_pointer (F%on_p, a2, (/ni,nj/));    a2 = GRID(ng)%on_p
    CALL c_f_pointer (F%omn, a2, (/ni,nj/));     a2 = GRID(ng)%omn
    CALL c_f_pointer (F%pmon_r, a2, (/ni,nj/));  a2 = GRID(ng)%pmon_r
    CALL c_f_pointer (F%pnom_r, a2, (/ni,nj/));  a2 = GRID(ng)%pnom_r
    CALL c_f_pointer (F%pmon_p, a2, (/ni,nj/));  a2 = GRID(ng)%pmon_p
    CALL c_f_pointer (F%pnom_p, a2, (/ni,nj/));  a2 = GRID(ng)%pnom_p
    CALL c_f_pointer (F%pmon_u, a2, (/ni,nj/));  a2 = GRID(ng)%pmon_u
    CALL c_f_pointer (F%pnom_u, a2, (/ni,nj/));  a2 = GRID(ng)%pnom_u
    CALL c_f_pointer (F%pmon_v, a2, (/ni,nj/));  a2 = GRID(ng)%pmon_v
    CALL c_f_pointer (F%pnom_v, a2, (/ni,nj/));  a2 = GRID(ng)%pnom_v
# ifdef BENCHMARK
    CALL c_f_pointer (F%dmde, a2, (/ni,nj/));    a2 = GRID(ng)%dmde
    CALL c_f_pointer (F%dndx, a2, (/ni,nj/));    a2 = GRID(ng)%dndx
# endif
  CASE (2)
    CALL set_scoord (ng)
    DO k = 1, NN
      scout(1+k) = SCALARS(ng)%sc_r(k)
      scout(NN+1+1+k) = SCALARS(ng)%Cs_r(k)
    END DO
    scout(1) = 0.0_dp; scout(NN+2) = 0.0_dp
    DO k = 0, NN
      scout(2*(NN+1)+1+k) = SCALARS(ng)%sc_w(k)
      scout(3*(NN+1)+1+k) = SCALARS(ng)%Cs_w(k)
    END DO
    scout(4*(NN+1)+1) = hc(ng)
  CASE (3)
    CALL c_f_pointer (F%h, a2, (/ni,nj/));        GRID(ng)%h = a2
    CALL c_f_pointer (F%Hz, a3, (/ni,nj,NN/));    GRID(ng)%Hz = a3
    CALL c_f_pointer (F%z_r, a3, (/ni,nj,NN/));   GRID(ng)%z_r = a3
    CALL c_f_pointer (F%z_w, a3, (/ni,nj,NN+1/)); GRID(ng)%z_w = a3
    !  ana_initial labels its min/max report with Vname(:,idTvar(itrc)), which the varinfo reader would fill
    IF (.NOT. allocated(idTvar)) THEN
      allocate ( idTvar(MT) )
      idTvar = 1
      Vname(1,1) = 'tracer'; Vname(2,1) = 'tracer'
    END IF
    CALL ana_initial (ng, tile, iNLM)
    CALL c_f_pointer (F%zeta, a3, (/ni,nj,3/));   a3 = OCEAN(ng)%zeta
    CALL c_f_pointer (F%ubar, a3, (/ni,nj,3/));   a3 = OCEAN(ng)%ubar
    CALL c_f_pointer (F%vbar, a3, (/ni,nj,3/));   a3 = OCEAN(ng)%vbar
    CALL c_f_pointer (F%u, a4, (/ni,nj,NN,2/));   a4 = OCEAN(ng)%u
    CALL c_f_pointer (F%v, a4, (/ni,nj,NN,2/));   a4 = OCEAN(ng)%v
    CALL c_f_pointer (F%t, a5, (/ni,nj,NN,3,NTT/)); a5 = OCEAN(ng)%t
  CASE (4)
# ifdef BENCHMARK
    CALL ana_winds (ng, tile, iNLM)
    CALL ana_tair (ng, tile, iNLM)
    CALL ana_pair (ng, tile, iNLM)
    CALL ana_humid (ng, tile, iNLM)
    CALL ana_rain (ng, tile, iNLM)
    CALL ana_cloud (ng, tile, iNLM)
    CALL c_f_pointer (F%Uwind, a2, (/ni,nj/));   a2 = FORCES(ng)%Uwind
    CALL c_f_pointer (F%Vwind, a2, (/ni,nj/));   a2 = FORCES(ng)%Vwind
    CALL c_f_pointer (F%Tair, a2, (/ni,nj/));    a2 = FORCES(ng)%Tair
    CALL c_f_pointer (F%Pair, a2, (/ni,nj/));    a2 = FORCES(ng)%Pair
    CALL c_f_pointer (F%Hair, a2, (/ni,nj/));    a2 = FORCES(ng)%Hair
    CALL c_f_pointer (F%rain, a2, (/ni,nj/));    a2 = FORCES(ng)%rain
    CALL c_f_pointer (F%cloud, a2, (/ni,nj/));   a2 = FORCES(ng)%cloud
# endif
# if defined UPWELLING || defined SEAMOUNT
    CALL ana_smflux (ng, tile, iNLM)
    CALL ana_stflux (ng, tile, iNLM, itemp)
    CALL c_f_pointer (F%sustr, a2, (/ni,nj/));   a2 = FORCES(ng)%sustr
    CALL c_f_pointer (F%svstr, a2, (/ni,nj/));   a2 = FORCES(ng)%svstr
    CALL c_f_pointer (F%stflux, a3, (/ni,nj,NTT/)); a3 = FORCES(ng)%stflux
# endif
# if defined UPWELLING && !defined REF_GLS
    !  ANA_VMIX: analytic vertical mixing coefficients on the z_w of F (ana_vmix.h)
    CALL c_f_pointer (F%z_w, a3, (/ni,nj,NN+1/)); GRID(ng)%z_w = a3
    CALL c_f_pointer (F%z_r, a3, (/ni,nj,NN/));   GRID(ng)%z_r = a3
    CALL c_f_pointer (F%h, a2, (/ni,nj/));        GRID(ng)%h = a2
    DO k = 1, INT(b%NAT)
      Akt_bak(k,ng) = p%Akt_bak(k)
    END DO
    Akv_bak(ng) = p%Akv_bak
    CALL ana_vmix (ng, tile, iNLM)
    CALL c_f_pointer (F%Akv, a3, (/ni,nj,NN+1/)); a3 = MIXING(ng)%Akv
    CALL c_f_pointer (F%Akt, a4, (/ni,nj,NN+1,INT(b%NAT)/)); a4 = MIXING(ng)%Akt
# endif
# ifdef BENCHMARK
  CASE (5)
    time_ref = 0.0_dp
    CALL ref_clock (time_ref)
    CALL c_f_pointer (F%lonr, a2, (/ni,nj/));    GRID(ng)%lonr = a2
    CALL c_f_pointer (F%latr, a2, (/ni,nj/));    GRID(ng)%latr = a2
    CALL c_f_pointer (F%Tair, a2, (/ni,nj/));    FORCES(ng)%Tair = a2
    CALL c_f_pointer (F%Hair, a2, (/ni,nj/));    FORCES(ng)%Hair = a2
    CALL c_f_pointer (F%cloud, a2, (/ni,nj/));   FORCES(ng)%cloud = a2
    CALL caldate (tdays(ng), yd_dp=scout(1), h_dp=scout(2))
    CALL ana_srflux (ng, tile, iNLM)
    CALL c_f_pointer (F%srflx, a2, (/ni,nj/));   a2 = FORCES(ng)%srflx
# endif
  CASE DEFAULT; rc = 2
  END SELECT
  CLOSE (78)
  stdout = saved_stdout
END FUNCTION ref_ana
#endif

!-----------------------------------------------------------------------
!  mpdata_adiff_tile (ROMS/Nonlinear/mpdata_adiff.F:38) on caller-held private
!  arrays: oHz, Ta, Ua, Va are (IminS:ImaxS,JminS:JmaxS,N), Wa is (..,..,0:N),
!  t3 is t(:,:,:,3,itrc) in module layout.
FUNCTION ref_mpdata_adiff (b, p, F, oHz, t3, Ta, Ua, Va, Wa) BIND(C, name='ref_mpdata_adiff') RESULT(rc)
  USE ref_wrap_types
  USE mod_param
  USE mod_scalars
  USE mod_ncparam
  USE mod_grid
  USE mod_ocean
  USE mpdata_adiff_mod, ONLY : mpdata_adiff_tile
  TYPE(bounds_t), INTENT(in) :: b
  TYPE(params_t), INTENT(in) :: p
  TYPE(fields_t), INTENT(in) :: F
  TYPE(c_ptr), VALUE :: oHz, t3, Ta, Ua, Va, Wa
  INTEGER(c_int) :: rc
  INTEGER :: ng, tile, LBi, UBi, LBj, UBj, ni, nj, NN, IminS, ImaxS, JminS, JmaxS, nis, njs
  REAL(c_double), POINTER :: a2(:,:), a3(:,:,:)
  REAL(c_double), POINTER :: poHz(:,:,:), pt3(:,:,:), pTa(:,:,:), pUa(:,:,:), pVa(:,:,:), pWa(:,:,:)
  ng = 1; tile = 0
  LBi = b%LBi; UBi = b%UBi; LBj = b%LBj; UBj = b%UBj
  ni = UBi-LBi+1; nj = UBj-LBj+1; NN = b%N
  rc = 0
  IF (LBi /= BOUNDS(ng)%LBi(0) .OR. UBi /= BOUNDS(ng)%UBi(0) .OR. LBj /= BOUNDS(ng)%LBj(0) .OR. UBj /= BOUNDS(ng)%UBj(0)) THEN
    rc = 1
    RETURN
  END IF
  dt(ng) = p%dt
  !  the generic LBC indices initialize_ncparam would set (mod_ncparam.F:1235-1236)
  isBu3d = isUvel
  isBv3d = isVvel
  !  (the routine asks whether the 3-D momentum's condition on the side is "closed"; p%lbc rows 4, 5 = u, v; 0 = the
  !  side's lbc_west / ... / lbc_north)
  LBC(iwest , isBu3d, ng)%closed = MERGE(p%lbc(4,1), p%lbc_west,  p%lbc(4,1) /= 0) == 1
  LBC(ieast , isBu3d, ng)%closed = MERGE(p%lbc(4,2), p%lbc_east,  p%lbc(4,2) /= 0) == 1
  LBC(isouth, isBu3d, ng)%closed = MERGE(p%lbc(4,3), p%lbc_south, p%lbc(4,3) /= 0) == 1
  LBC(inorth, isBu3d, ng)%closed = MERGE(p%lbc(4,4), p%lbc_north, p%lbc(4,4) /= 0) == 1
  LBC(iwest , isBv3d, ng)%closed = MERGE(p%lbc(5,1), p%lbc_west,  p%lbc(5,1) /= 0) == 1
  LBC(ieast , isBv3d, ng)%closed = MERGE(p%lbc(5,2), p%lbc_east,  p%lbc(5,2) /= 0) == 1
  LBC(isouth, isBv3d, ng)%closed = MERGE(p%lbc(5,3), p%lbc_south, p%lbc(5,3) /= 0) == 1
  LBC(inorth, isBv3d, ng)%closed = MERGE(p%lbc(5,4), p%lbc_north, p%lbc(5,4) /= 0) == 1
  CALL c_f_pointer (F%pm, a2, (/ni,nj/));       GRID(ng)%pm = a2
  CALL c_f_pointer (F%pn, a2, (/ni,nj/));       GRID(ng)%pn = a2
  CALL c_f_pointer (F%omn, a2, (/ni,nj/));      GRID(ng)%omn = a2
  CALL c_f_pointer (F%om_u, a2, (/ni,nj/));     GRID(ng)%om_u = a2
  CALL c_f_pointer (F%on_v, a2, (/ni,nj/));     GRID(ng)%on_v = a2
  CALL c_f_pointer (F%z_r, a3, (/ni,nj,NN/));   GRID(ng)%z_r = a3
  CALL c_f_pointer (F%Huon, a3, (/ni,nj,NN/));  GRID(ng)%Huon = a3
  CALL c_f_pointer (F%Hvom, a3, (/ni,nj,NN/));  GRID(ng)%Hvom = a3
  CALL c_f_pointer (F%W, a3, (/ni,nj,NN+1/));   OCEAN(ng)%W = a3
  IminS = BOUNDS(ng)%Istr(tile)-3; ImaxS = BOUNDS(ng)%Iend(tile)+3
  JminS = BOUNDS(ng)%Jstr(tile)-3; JmaxS = BOUNDS(ng)%Jend(tile)+3
  nis = ImaxS-IminS+1; njs = JmaxS-JminS+1
  CALL c_f_pointer (oHz, poHz, (/nis,njs,NN/))
  CALL c_f_pointer (t3,  pt3,  (/ni,nj,NN/))
  CALL c_f_pointer (Ta,  pTa,  (/nis,njs,NN/))
  CALL c_f_pointer (Ua,  pUa,  (/nis,njs,NN/))
  CALL c_f_pointer (Va,  pVa,  (/nis,njs,NN/))
  CALL c_f_pointer (Wa,  pWa,  (/nis,njs,NN+1/))
#ifdef MASKING
  CALL c_f_pointer (F%rmask, a2, (/ni,nj/));    GRID(ng)%rmask = a2
  CALL c_f_pointer (F%umask, a2, (/ni,nj/));    GRID(ng)%umask = a2
  CALL c_f_pointer (F%vmask, a2, (/ni,nj/));    GRID(ng)%vmask = a2
#endif
#ifdef WET_DRY
  CALL c_f_pointer (F%rmask_wet, a2, (/ni,nj/)); GRID(ng)%rmask_wet = a2
  CALL c_f_pointer (F%umask_wet, a2, (/ni,nj/)); GRID(ng)%umask_wet = a2
  CALL c_f_pointer (F%vmask_wet, a2, (/ni,nj/)); GRID(ng)%vmask_wet = a2
  CALL c_f_pointer (F%pmask_wet, a2, (/ni,nj/)); GRID(ng)%pmask_wet = a2
  Dcrit(ng) = p%Dcrit
#endif
  CALL mpdata_adiff_tile (ng, tile, LBi, UBi, LBj, UBj, IminS, ImaxS, JminS, JmaxS,   &
#ifdef MASKING
 &                        GRID(ng)%rmask, GRID(ng)%umask, GRID(ng)%vmask, &
#endif
#ifdef WET_DRY
 &                        GRID(ng)%rmask_wet, GRID(ng)%umask_wet, GRID(ng)%vmask_wet, &
#endif
 &                        GRID(ng)%pm, GRID(ng)%pn, GRID(ng)%omn, GRID(ng)%om_u, GRID(ng)%on_v, &
 &                        GRID(ng)%z_r, poHz, GRID(ng)%Huon, GRID(ng)%Hvom, OCEAN(ng)%W, &
 &                        pt3, pTa, pUa, pVa, pWa)
END FUNCTION ref_mpdata_adiff

!-----------------------------------------------------------------------
!  The lateral boundary-condition routines of the reference on the S/N edges, through their own _tile
!  procedures: kind 1 zetabc_tile, 2 u2dbc_tile, 3 v2dbc_tile, 4 u3dbc_tile, 5 v3dbc_tile, 6 t3dbc_tile (tracer
!  itrc); 7 ini_zeta, 8 ini_fields (ini_fields.F:780, :27 -- the first-step initialisation of main3d.F:269-283,
!  which applies those conditions).  LBC(:, isFsur..isTvar, ng) is filled from p%lbc (codes of enum roms_lbc), the BOUNDARY(ng)%*_south /
!  *_north vectors from the *_bry fields (the value of a boundary point sits at that point, roms_fields.def).
FUNCTION ref_bc (kind, b, p, s, F, nout, itrc) BIND(C, name='ref_bc') RESULT(rc)
  USE ref_wrap_types
  USE mod_param
  USE mod_scalars
  USE mod_stepping
  USE mod_ncparam
  USE mod_grid
  USE mod_ocean
  USE mod_boundary
  USE mod_forces
  USE zetabc_mod, ONLY : zetabc_tile
  USE u2dbc_mod,  ONLY : u2dbc_tile
  USE v2dbc_mod,  ONLY : v2dbc_tile
  USE u3dbc_mod,  ONLY : u3dbc_tile
  USE v3dbc_mod,  ONLY : v3dbc_tile
  USE t3dbc_mod,  ONLY : t3dbc_tile
  USE ini_fields_mod, ONLY : ini_fields, ini_zeta
  USE mod_coupling
  INTEGER(c_int), VALUE :: kind, nout, itrc
  TYPE(bounds_t), INTENT(in) :: b
  TYPE(params_t), INTENT(in) :: p
  TYPE(stepidx_t), INTENT(in) :: s
  TYPE(fields_t), INTENT(in) :: F
  INTEGER(c_int) :: rc
  INTEGER :: ng, tile, LBi, UBi, LBj, UBj, ni, nj, NN, NTT, sd, v, code, side(4), ivar, i, k, it, eff4(6,4)
  INTEGER :: IminS, ImaxS, JminS, JmaxS, Jstr, Jend, Istr, Iend
  REAL(c_double), POINTER :: a2(:,:), a3(:,:,:), a4(:,:,:,:), a5(:,:,:,:,:)
  ng = 1; tile = 0
  LBi = b%LBi; UBi = b%UBi; LBj = b%LBj; UBj = b%UBj
  ni = UBi-LBi+1; nj = UBj-LBj+1; NN = b%N; NTT = b%NT
  Jstr = b%Jstr; Jend = b%Jend; Istr = b%Istr; Iend = b%Iend
  IminS = b%Istr-3; ImaxS = b%Iend+3; JminS = b%Jstr-3; JmaxS = b%Jend+3
  rc = 0
  IF (.NOT. have_boundary) THEN
    CALL allocate_boundary (ng)
    have_boundary = .TRUE.
  END IF
  IF (.NOT. allocated(isTvar)) THEN
    allocate ( isTvar(MT) )
    DO it = 1, MT
      isTvar(it) = 5 + it                     ! mod_ncparam.F:1196-1203
    END DO
  END IF
  nstp(ng) = s%nstp; nnew(ng) = s%nnew; nrhs(ng) = s%nrhs
  kstp(ng) = s%kstp; krhs(ng) = s%krhs; knew(ng) = s%knew
  iif(ng) = s%iif; iic(ng) = s%iic; ntfirst(ng) = s%ntfirst
  PREDICTOR_2D_STEP(ng) = s%predictor /= 0
  dt(ng) = p%dt; dtfast(ng) = p%dtfast
  g = p%g; rho0 = p%rho0; gamma2(ng) = p%gamma2
  Co = 1.0_c_double/(2.0_c_double+SQRT(2.0_c_double))      ! as initialize_scalars sets it (mod_scalars.F:4175)
  PerfectRST(ng) = .FALSE.
  ! LBC(side, variable): columns 1..4 of p%lbc = west, east, south, north (enum roms_lbc_side + 1); code 0 = the
  ! side's lbc_west / lbc_east / lbc_south / lbc_north, whose value 0 means periodic
  side(1) = iwest; side(2) = ieast; side(3) = isouth; side(4) = inorth
  DO sd = 1, 4
    DO v = 1, 6
      code = p%lbc(v, sd)
      IF (code == 0) THEN
        IF (sd == 1) code = p%lbc_west
        IF (sd == 2) code = p%lbc_east
        IF (sd == 3) code = p%lbc_south
        IF (sd == 4) code = p%lbc_north
      END IF
      DO it = 1, MERGE(NTT, 1, v == 6)
        ivar = v
        IF (v == 6) ivar = isTvar(it)
        LBC(side(sd), ivar, ng)%periodic = code == 0
        LBC(side(sd), ivar, ng)%closed = code == 1
        LBC(side(sd), ivar, ng)%gradient = code == 2
        LBC(side(sd), ivar, ng)%clamped = code == 3
        LBC(side(sd), ivar, ng)%Chapman_implicit = code == 4
        LBC(side(sd), ivar, ng)%Flather = code == 5
        LBC(side(sd), ivar, ng)%radiation = code == 6 .OR. code == 7
        LBC(side(sd), ivar, ng)%Chapman_explicit = code == 8
        LBC(side(sd), ivar, ng)%nudging = code == 7                 ! "RadNud"
        LBC(side(sd), ivar, ng)%nested = .FALSE.
        LBC(side(sd), ivar, ng)%reduced = code == 10
        LBC(side(sd), ivar, ng)%Shchepetkin = code == 9
        LBC(side(sd), ivar, ng)%acquire = .FALSE.
      END DO
      eff4(v, sd) = code
    END DO
  END DO
  !  boundary data of the free surface are "acquired" on a side whose free-surface condition is clamped or nudged, or
  !  whose ubar / vbar condition is Flather or Shchepetkin (inp_decode.F:1620-1655; no FSOBC_REDUCED): the reduced-
  !  physics condition then takes its pressure gradient from them (u2dbc_im.F:395-405)
  DO sd = 1, 4
    LBC(side(sd), isFsur, ng)%acquire = eff4(1,sd) == 3 .OR. eff4(1,sd) == 7 .OR. eff4(2,sd) == 5 .OR. eff4(2,sd) == 9 &
 &                                      .OR. eff4(3,sd) == 5 .OR. eff4(3,sd) == 9
  END DO
  ! nudging coefficients of RadNud edges (constant ones: no climatology nudging coefficients)
  LnudgeM2CLM(ng) = .FALSE.; LnudgeM3CLM(ng) = .FALSE.
  DO it = 1, NTT
    LnudgeTCLM(it,ng) = .FALSE.
  END DO
  DO sd = 1, 4
    FSobc_out(ng,side(sd)) = p%obc_out(1,sd); FSobc_in(ng,side(sd)) = p%obc_in(1,sd)
    ! (the reference has ONE pair for both components of the 2-D / 3-D momentum; the library's table has a column per
    ! variable: take the column of the routine that is being called)
    M2obc_out(ng,side(sd)) = p%obc_out(MERGE(3, 2, kind == 3),sd); M2obc_in(ng,side(sd)) = p%obc_in(MERGE(3, 2, kind == 3),sd)
    M3obc_out(ng,side(sd)) = p%obc_out(MERGE(5, 4, kind == 5),sd); M3obc_in(ng,side(sd)) = p%obc_in(MERGE(5, 4, kind == 5),sd)
    DO it = 1, NTT
      Tobc_out(it,ng,side(sd)) = p%obc_out(6,sd); Tobc_in(it,ng,side(sd)) = p%obc_in(6,sd)
    END DO
  END DO
  ! ---- grid, masks, state ----
  CALL c_f_pointer (F%h, a2, (/ni,nj/));        GRID(ng)%h = a2
  CALL c_f_pointer (F%f, a2, (/ni,nj/));        GRID(ng)%f = a2
  CALL c_f_pointer (F%pm, a2, (/ni,nj/));       GRID(ng)%pm = a2
  CALL c_f_pointer (F%pn, a2, (/ni,nj/));       GRID(ng)%pn = a2
  CALL c_f_pointer (F%om_u, a2, (/ni,nj/));     GRID(ng)%om_u = a2
  CALL c_f_pointer (F%on_u, a2, (/ni,nj/));     GRID(ng)%on_u = a2
  CALL c_f_pointer (F%om_v, a2, (/ni,nj/));     GRID(ng)%om_v = a2
  CALL c_f_pointer (F%on_v, a2, (/ni,nj/));     GRID(ng)%on_v = a2
#ifdef MASKING
  CALL c_f_pointer (F%rmask, a2, (/ni,nj/));    GRID(ng)%rmask = a2
  CALL c_f_pointer (F%umask, a2, (/ni,nj/));    GRID(ng)%umask = a2
  CALL c_f_pointer (F%vmask, a2, (/ni,nj/));    GRID(ng)%vmask = a2
#endif
#ifdef WET_DRY
  CALL c_f_pointer (F%rmask_wet, a2, (/ni,nj/)); GRID(ng)%rmask_wet = a2
  CALL c_f_pointer (F%umask_wet, a2, (/ni,nj/)); GRID(ng)%umask_wet = a2
  CALL c_f_pointer (F%vmask_wet, a2, (/ni,nj/)); GRID(ng)%vmask_wet = a2
  CALL c_f_pointer (F%pmask_wet, a2, (/ni,nj/)); GRID(ng)%pmask_wet = a2
  Dcrit(ng) = p%Dcrit
#endif
  CALL c_f_pointer (F%zeta, a3, (/ni,nj,3/));   OCEAN(ng)%zeta = a3
  CALL c_f_pointer (F%ubar, a3, (/ni,nj,3/));   OCEAN(ng)%ubar = a3
  CALL c_f_pointer (F%vbar, a3, (/ni,nj,3/));   OCEAN(ng)%vbar = a3
  CALL c_f_pointer (F%u, a4, (/ni,nj,NN,2/));   OCEAN(ng)%u = a4
  CALL c_f_pointer (F%v, a4, (/ni,nj,NN,2/));   OCEAN(ng)%v = a4
  CALL c_f_pointer (F%t, a5, (/ni,nj,NN,3,NTT/)); OCEAN(ng)%t = a5
  CALL c_f_pointer (F%Hz, a3, (/ni,nj,NN/));    GRID(ng)%Hz = a3
  CALL c_f_pointer (F%Zt_avg1, a2, (/ni,nj/));  COUPLING(ng)%Zt_avg1 = a2
#ifdef ATM_PRESS
  CALL c_f_pointer (F%Pair, a2, (/ni,nj/));     FORCES(ng)%Pair = a2           ! (PRESS_COMPENSATE in the Flather value)
#endif
  CALL c_f_pointer (F%sustr, a2, (/ni,nj/));    FORCES(ng)%sustr = a2          ! (the reduced-physics condition)
  CALL c_f_pointer (F%svstr, a2, (/ni,nj/));    FORCES(ng)%svstr = a2
  CALL c_f_pointer (F%bustr, a2, (/ni,nj/));    FORCES(ng)%bustr = a2
  CALL c_f_pointer (F%bvstr, a2, (/ni,nj/));    FORCES(ng)%bvstr = a2
  ! ---- boundary data: edge vectors from the rows of the *_bry fields ----
  IF (.NOT. associated(BOUNDARY(ng)%zeta_south)) THEN
    allocate ( BOUNDARY(ng)%zeta_south(LBi:UBi), BOUNDARY(ng)%zeta_north(LBi:UBi) )
    allocate ( BOUNDARY(ng)%ubar_south(LBi:UBi), BOUNDARY(ng)%ubar_north(LBi:UBi) )
    allocate ( BOUNDARY(ng)%vbar_south(LBi:UBi), BOUNDARY(ng)%vbar_north(LBi:UBi) )
    allocate ( BOUNDARY(ng)%u_south(LBi:UBi,NN), BOUNDARY(ng)%u_north(LBi:UBi,NN) )
    allocate ( BOUNDARY(ng)%v_south(LBi:UBi,NN), BOUNDARY(ng)%v_north(LBi:UBi,NN) )
    allocate ( BOUNDARY(ng)%t_south(LBi:UBi,NN,NTT), BOUNDARY(ng)%t_north(LBi:UBi,NN,NTT) )
    allocate ( BOUNDARY(ng)%zeta_west(LBj:UBj), BOUNDARY(ng)%zeta_east(LBj:UBj) )
    allocate ( BOUNDARY(ng)%ubar_west(LBj:UBj), BOUNDARY(ng)%ubar_east(LBj:UBj) )
    allocate ( BOUNDARY(ng)%vbar_west(LBj:UBj), BOUNDARY(ng)%vbar_east(LBj:UBj) )
    allocate ( BOUNDARY(ng)%u_west(LBj:UBj,NN), BOUNDARY(ng)%u_east(LBj:UBj,NN) )
    allocate ( BOUNDARY(ng)%v_west(LBj:UBj,NN), BOUNDARY(ng)%v_east(LBj:UBj,NN) )
    allocate ( BOUNDARY(ng)%t_west(LBj:UBj,NN,NTT), BOUNDARY(ng)%t_east(LBj:UBj,NN,NTT) )
  END IF
  CALL c_f_pointer (F%zeta_bry, a2, (/ni,nj/))
  BOUNDARY(ng)%zeta_south(LBi:UBi) = a2(:, Jstr-1-LBj+1); BOUNDARY(ng)%zeta_north(LBi:UBi) = a2(:, Jend+1-LBj+1)
  CALL c_f_pointer (F%ubar_bry, a2, (/ni,nj/))
  BOUNDARY(ng)%ubar_south(LBi:UBi) = a2(:, Jstr-1-LBj+1); BOUNDARY(ng)%ubar_north(LBi:UBi) = a2(:, Jend+1-LBj+1)
  CALL c_f_pointer (F%vbar_bry, a2, (/ni,nj/))
  BOUNDARY(ng)%vbar_south(LBi:UBi) = a2(:, Jstr-LBj+1);   BOUNDARY(ng)%vbar_north(LBi:UBi) = a2(:, Jend+1-LBj+1)
  CALL c_f_pointer (F%u_bry, a3, (/ni,nj,NN/))
  BOUNDARY(ng)%u_south(LBi:UBi,1:NN) = a3(:, Jstr-1-LBj+1, :); BOUNDARY(ng)%u_north(LBi:UBi,1:NN) = a3(:, Jend+1-LBj+1, :)
  CALL c_f_pointer (F%v_bry, a3, (/ni,nj,NN/))
  BOUNDARY(ng)%v_south(LBi:UBi,1:NN) = a3(:, Jstr-LBj+1, :);   BOUNDARY(ng)%v_north(LBi:UBi,1:NN) = a3(:, Jend+1-LBj+1, :)
  CALL c_f_pointer (F%t_bry, a4, (/ni,nj,NN,NTT/))
  BOUNDARY(ng)%t_south(LBi:UBi,1:NN,1:NTT) = a4(:, Jstr-1-LBj+1, :, :)
  BOUNDARY(ng)%t_north(LBi:UBi,1:NN,1:NTT) = a4(:, Jend+1-LBj+1, :, :)
  ! western / eastern edge vectors: the boundary column of each field (rho- and v-type: Istr-1 / Iend+1, u-type:
  ! Istr / Iend+1)
  CALL c_f_pointer (F%zeta_bry, a2, (/ni,nj/))
  BOUNDARY(ng)%zeta_west(LBj:UBj) = a2(Istr-1-LBi+1, :); BOUNDARY(ng)%zeta_east(LBj:UBj) = a2(Iend+1-LBi+1, :)
  CALL c_f_pointer (F%ubar_bry, a2, (/ni,nj/))
  BOUNDARY(ng)%ubar_west(LBj:UBj) = a2(Istr-LBi+1, :);   BOUNDARY(ng)%ubar_east(LBj:UBj) = a2(Iend+1-LBi+1, :)
  CALL c_f_pointer (F%vbar_bry, a2, (/ni,nj/))
  BOUNDARY(ng)%vbar_west(LBj:UBj) = a2(Istr-1-LBi+1, :); BOUNDARY(ng)%vbar_east(LBj:UBj) = a2(Iend+1-LBi+1, :)
  CALL c_f_pointer (F%u_bry, a3, (/ni,nj,NN/))
  BOUNDARY(ng)%u_west(LBj:UBj,1:NN) = a3(Istr-LBi+1, :, :);   BOUNDARY(ng)%u_east(LBj:UBj,1:NN) = a3(Iend+1-LBi+1, :, :)
  CALL c_f_pointer (F%v_bry, a3, (/ni,nj,NN/))
  BOUNDARY(ng)%v_west(LBj:UBj,1:NN) = a3(Istr-1-LBi+1, :, :); BOUNDARY(ng)%v_east(LBj:UBj,1:NN) = a3(Iend+1-LBi+1, :, :)
  CALL c_f_pointer (F%t_bry, a4, (/ni,nj,NN,NTT/))
  BOUNDARY(ng)%t_west(LBj:UBj,1:NN,1:NTT) = a4(Istr-1-LBi+1, :, :, :)
  BOUNDARY(ng)%t_east(LBj:UBj,1:NN,1:NTT) = a4(Iend+1-LBi+1, :, :, :)
  ! ---- the reference procedure ----
  SELECT CASE (kind)
  CASE (1); CALL zetabc_tile (ng, tile, LBi, UBi, LBj, UBj, IminS, ImaxS, JminS, JmaxS, s%krhs, s%kstp, nout, OCEAN(ng)%zeta)
  CASE (2); CALL u2dbc_tile (ng, tile, LBi, UBi, LBj, UBj, IminS, ImaxS, JminS, JmaxS, s%krhs, s%kstp, nout,       &
 &                           OCEAN(ng)%ubar, OCEAN(ng)%vbar, OCEAN(ng)%zeta)
  CASE (3); CALL v2dbc_tile (ng, tile, LBi, UBi, LBj, UBj, IminS, ImaxS, JminS, JmaxS, s%krhs, s%kstp, nout,       &
 &                           OCEAN(ng)%ubar, OCEAN(ng)%vbar, OCEAN(ng)%zeta)
  CASE (4); CALL u3dbc_tile (ng, tile, LBi, UBi, LBj, UBj, NN, IminS, ImaxS, JminS, JmaxS, s%nstp, nout, OCEAN(ng)%u)
  CASE (5); CALL v3dbc_tile (ng, tile, LBi, UBi, LBj, UBj, NN, IminS, ImaxS, JminS, JmaxS, s%nstp, nout, OCEAN(ng)%v)
  CASE (6); CALL t3dbc_tile (ng, tile, itrc, 0, LBi, UBi, LBj, UBj, NN, NTT, IminS, ImaxS, JminS, JmaxS,        &
 &                           s%nstp, nout, OCEAN(ng)%t)
  CASE (7); CALL ini_zeta (ng, tile, iNLM)                 ! main3d.F:275
  CASE (8); CALL ini_fields (ng, tile, iNLM)               ! main3d.F:282
  CASE DEFAULT; rc = 2
  END SELECT
  ! ---- copy out ----
  CALL c_f_pointer (F%Zt_avg1, a2, (/ni,nj/));  a2 = COUPLING(ng)%Zt_avg1
  CALL c_f_pointer (F%zeta, a3, (/ni,nj,3/));   a3 = OCEAN(ng)%zeta
  CALL c_f_pointer (F%ubar, a3, (/ni,nj,3/));   a3 = OCEAN(ng)%ubar
  CALL c_f_pointer (F%vbar, a3, (/ni,nj,3/));   a3 = OCEAN(ng)%vbar
  CALL c_f_pointer (F%u, a4, (/ni,nj,NN,2/));   a4 = OCEAN(ng)%u
  CALL c_f_pointer (F%v, a4, (/ni,nj,NN,2/));   a4 = OCEAN(ng)%v
  CALL c_f_pointer (F%t, a5, (/ni,nj,NN,3,NTT/)); a5 = OCEAN(ng)%t
END FUNCTION ref_bc

!-----------------------------------------------------------------------
!  GLS_MIXING (builds with -DREF_GLS: ref_headers/upwelling_gls.h, benchmark_gls.h): kernel 1 = gls_prestep
!  (gls_prestep.F:23), 2 = gls_corstep (gls_corstep.F:27), as main3d.F:567 / :793 call them.  The closure
!  parameters of roms_*.in (GLS_P ... GLS_SIGP, AKK_BAK, AKP_BAK, ZOS) come from the params block; the stability
!  constants are what initialize_scalars (ref_setup) computed.  LBC(:,isMtke,ng) = the tracers' table (closed or
!  gradient: tkebc_im.F treats both alike).  With -DREF_MY25 as well (the _my25 headers of ref_headers) the two kernels are
!  my25_prestep (my25_prestep.F:23) and my25_corstep (my25_corstep.F:27); MIXING(ng)%Akp does not exist then.
#ifdef REF_GLS
FUNCTION ref_gls (kernel, b, p, s, F) BIND(C, name='ref_gls') RESULT(rc)
  USE ref_wrap_types
  USE mod_param
  USE mod_scalars
  USE mod_ncparam
  USE mod_stepping
  USE mod_grid
  USE mod_ocean
  USE mod_mixing
  USE mod_forces
  USE mod_boundary, ONLY : allocate_boundary
# ifdef REF_MY25
  USE my25_prestep_mod, ONLY : my25_prestep
  USE my25_corstep_mod, ONLY : my25_corstep
# else
  USE gls_prestep_mod, ONLY : gls_prestep
  USE gls_corstep_mod, ONLY : gls_corstep
# endif
  INTEGER(c_int), VALUE :: kernel
  TYPE(bounds_t), INTENT(in) :: b
  TYPE(params_t), INTENT(in) :: p
  TYPE(stepidx_t), INTENT(in) :: s
  TYPE(fields_t), INTENT(in) :: F
  INTEGER(c_int) :: rc
  INTEGER :: ng, tile, LBi, UBi, LBj, UBj, ni, nj, NN, NTT, sd, code, it
  INTEGER :: side4(4)
  REAL(c_double), POINTER :: a2(:,:), a3(:,:,:), a4(:,:,:,:)
  ng = 1; tile = 0
  LBi = b%LBi; UBi = b%UBi; LBj = b%LBj; UBj = b%UBj
  ni = UBi-LBi+1; nj = UBj-LBj+1; NN = b%N; NTT = b%NT
  rc = 0
  IF (.NOT. have_boundary) THEN
    CALL allocate_boundary (ng)
    have_boundary = .TRUE.
  END IF
  nstp(ng) = s%nstp; nnew(ng) = s%nnew; nrhs(ng) = s%nrhs
  iic(ng) = s%iic; ntfirst(ng) = s%ntfirst
  dt(ng) = p%dt; g = p%g; rho0 = p%rho0
  gls_p(ng) = p%gls_p; gls_m(ng) = p%gls_m; gls_n(ng) = p%gls_n; gls_cmu0(ng) = p%gls_cmu0
  gls_c1(ng) = p%gls_c1; gls_c2(ng) = p%gls_c2; gls_c3m(ng) = p%gls_c3m; gls_c3p(ng) = p%gls_c3p
  gls_sigk(ng) = p%gls_sigk; gls_sigp(ng) = p%gls_sigp; gls_Kmin(ng) = p%gls_Kmin; gls_Pmin(ng) = p%gls_Pmin
  Akk_bak(ng) = p%Akk_bak; Akp_bak(ng) = p%Akp_bak; Akv_bak(ng) = p%Akv_bak; Zos(ng) = p%Zos
  DO it = 1, INT(b%NAT)
    Akt_bak(it,ng) = p%Akt_bak(it)
  END DO
  !  LBC(:,isMtke,ng): index 6 (= isTvar(1), free in this build: initialize_ncparam has not run)
  isMtke = 6
  side4 = (/ iwest, ieast, isouth, inorth /)
  DO sd = 1, 4
    code = p%lbc(6, sd)
    IF (code == 0) THEN
      SELECT CASE (sd)
      CASE (1); code = p%lbc_west
      CASE (2); code = p%lbc_east
      CASE (3); code = p%lbc_south
      CASE (4); code = p%lbc_north
      END SELECT
    END IF
    LBC(side4(sd), isMtke, ng)%closed = code == 1
    LBC(side4(sd), isMtke, ng)%gradient = code == 2
    LBC(side4(sd), isMtke, ng)%radiation = .FALSE.
    LBC(side4(sd), isMtke, ng)%periodic = code == 0
  END DO
  CALL c_f_pointer (F%pm, a2, (/ni,nj/));       GRID(ng)%pm = a2
  CALL c_f_pointer (F%pn, a2, (/ni,nj/));       GRID(ng)%pn = a2
  CALL c_f_pointer (F%Hz, a3, (/ni,nj,NN/));    GRID(ng)%Hz = a3
  CALL c_f_pointer (F%Huon, a3, (/ni,nj,NN/));  GRID(ng)%Huon = a3
  CALL c_f_pointer (F%Hvom, a3, (/ni,nj,NN/));  GRID(ng)%Hvom = a3
  CALL c_f_pointer (F%z_r, a3, (/ni,nj,NN/));   GRID(ng)%z_r = a3
  CALL c_f_pointer (F%z_w, a3, (/ni,nj,NN+1/)); GRID(ng)%z_w = a3
# ifndef REF_MY25
  CALL c_f_pointer (F%ZoBot, a2, (/ni,nj/));    GRID(ng)%ZoBot = a2
# endif
#ifdef MASKING
  CALL c_f_pointer (F%rmask, a2, (/ni,nj/));    GRID(ng)%rmask = a2
  CALL c_f_pointer (F%umask, a2, (/ni,nj/));    GRID(ng)%umask = a2
  CALL c_f_pointer (F%vmask, a2, (/ni,nj/));    GRID(ng)%vmask = a2
#endif
#ifdef WET_DRY
  CALL c_f_pointer (F%rmask_wet, a2, (/ni,nj/)); GRID(ng)%rmask_wet = a2
  CALL c_f_pointer (F%umask_wet, a2, (/ni,nj/)); GRID(ng)%umask_wet = a2
  CALL c_f_pointer (F%vmask_wet, a2, (/ni,nj/)); GRID(ng)%vmask_wet = a2
  CALL c_f_pointer (F%pmask_wet, a2, (/ni,nj/)); GRID(ng)%pmask_wet = a2
  Dcrit(ng) = p%Dcrit
#endif
  CALL c_f_pointer (F%u, a4, (/ni,nj,NN,2/));   OCEAN(ng)%u = a4
  CALL c_f_pointer (F%v, a4, (/ni,nj,NN,2/));   OCEAN(ng)%v = a4
  CALL c_f_pointer (F%W, a3, (/ni,nj,NN+1/));   OCEAN(ng)%W = a3
  CALL c_f_pointer (F%bustr, a2, (/ni,nj/));    FORCES(ng)%bustr = a2
  CALL c_f_pointer (F%bvstr, a2, (/ni,nj/));    FORCES(ng)%bvstr = a2
  CALL c_f_pointer (F%sustr, a2, (/ni,nj/));    FORCES(ng)%sustr = a2
  CALL c_f_pointer (F%svstr, a2, (/ni,nj/));    FORCES(ng)%svstr = a2
  CALL c_f_pointer (F%bvf, a3, (/ni,nj,NN+1/)); MIXING(ng)%bvf = a3
  CALL c_f_pointer (F%Akv, a3, (/ni,nj,NN+1/)); MIXING(ng)%Akv = a3
  CALL c_f_pointer (F%Akt, a4, (/ni,nj,NN+1,INT(b%NAT)/)); MIXING(ng)%Akt = a4
  CALL c_f_pointer (F%Akk, a3, (/ni,nj,NN+1/)); MIXING(ng)%Akk = a3
# ifndef REF_MY25
  CALL c_f_pointer (F%Akp, a3, (/ni,nj,NN+1/)); MIXING(ng)%Akp = a3
# endif
  CALL c_f_pointer (F%Lscale, a3, (/ni,nj,NN+1/)); MIXING(ng)%Lscale = a3
  CALL c_f_pointer (F%tke, a4, (/ni,nj,NN+1,3/)); MIXING(ng)%tke = a4
  CALL c_f_pointer (F%gls, a4, (/ni,nj,NN+1,3/)); MIXING(ng)%gls = a4
  SELECT CASE (kernel)
# ifdef REF_MY25
  CASE (1); CALL my25_prestep (ng, tile)
  CASE (2); CALL my25_corstep (ng, tile)
# else
  CASE (1); CALL gls_prestep (ng, tile)
  CASE (2); CALL gls_corstep (ng, tile)
# endif
  CASE DEFAULT; rc = 2
  END SELECT
  CALL c_f_pointer (F%tke, a4, (/ni,nj,NN+1,3/)); a4 = MIXING(ng)%tke
  CALL c_f_pointer (F%gls, a4, (/ni,nj,NN+1,3/)); a4 = MIXING(ng)%gls
  CALL c_f_pointer (F%Akv, a3, (/ni,nj,NN+1/)); a3 = MIXING(ng)%Akv
  CALL c_f_pointer (F%Akt, a4, (/ni,nj,NN+1,INT(b%NAT)/)); a4 = MIXING(ng)%Akt
  CALL c_f_pointer (F%Akk, a3, (/ni,nj,NN+1/)); a3 = MIXING(ng)%Akk
# ifndef REF_MY25
  CALL c_f_pointer (F%Akp, a3, (/ni,nj,NN+1/)); a3 = MIXING(ng)%Akp
# endif
  CALL c_f_pointer (F%Lscale, a3, (/ni,nj,NN+1/)); a3 = MIXING(ng)%Lscale
END FUNCTION ref_gls
#endif
