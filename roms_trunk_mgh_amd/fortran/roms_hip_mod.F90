!=======================================================================
!  roms_hip_mod -- thin ISO_C_BINDING layer between the unchanged ROMS
!  Fortran driver (ROMS/Nonlinear/main3d.F) and libroms_hip.so.
!
!  The reference's seam is the module-procedure name: main3d does
!  `USE step3d_t_mod, ONLY : step3d_t` and `CALL step3d_t (ng, tile)`
!  (main3d.F:94-124, :814).  A maintainer keeps main3d.F as it is and
!  replaces the BODY of each wrapper `X(ng,tile)` by a call into this
!  module (see INTEGRATION.md for the per-file edits), e.g.
!
!      SUBROUTINE step3d_t (ng, tile)          ! step3d_t.F:40
!        USE roms_hip_mod
!        integer, intent(in) :: ng, tile
!        TYPE(roms_step_idx_t) :: s
!        s = roms_hip_make_idx (iic(ng), ntfirst(ng), nstp(ng), nnew(ng), nrhs(ng), kstp(ng), krhs(ng),   &
!     &                         knew(ng), iif(ng), PREDICTOR_2D_STEP(ng))
!        CALL roms_hip_status (roms_hip_step3d_t (s), exit_flag)
!      END SUBROUTINE step3d_t
!
!  Everything here is interface + glue: no arithmetic of the hot path is
!  done on the host.
!=======================================================================
MODULE roms_hip_mod
  USE, INTRINSIC :: iso_c_binding
  IMPLICIT NONE
  PRIVATE

  !  mirrors `roms_step_idx_t` of include/roms_hip.h
  TYPE, BIND(C), PUBLIC :: roms_step_idx_t
    INTEGER(c_int) :: iic, ntfirst
    INTEGER(c_int) :: nstp, nnew, nrhs
    INTEGER(c_int) :: kstp, krhs, knew
    INTEGER(c_int) :: iif, predictor_2d_step
  END TYPE roms_step_idx_t

  !  mirror `roms_bounds_t` and `roms_params_t` of include/roms_hip.h field for field (sizes checked against
  !  roms_abi_sizeof in tests/test_fortran_shim.py): fill one of each from BOUNDS(ng)%...(tile), DOMAIN(ng)%*_Edge(tile)
  !  and mod_scalars, declare it TARGET and hand c_loc of it to roms_hip_set_bounds / roms_hip_set_params
  TYPE, BIND(C), PUBLIC :: roms_bounds_t
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
  END TYPE roms_bounds_t
  TYPE, BIND(C), PUBLIC :: roms_params_t
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
    INTEGER(c_int) :: ts_dif4, uv_vis4            ! TS_DIF4, UV_VIS4 (biharmonic mixing)
    INTEGER(c_int) :: mix_iso_ts, radiation_2d    ! MIX_ISO_TS, RADIATION_2D
    REAL(c_double) :: Cdb_min, Cdb_max            ! UV_LOGDRAG limits
    !  GLS_MIXING (gls_prestep.F, gls_corstep.F): switch, stability functions (0 Galperin, 1 KANTHA_CLAYSON,
    !  2 CANUTO_A, 3 CANUTO_B), N2S2_HORAVG, RI_SPLINES; the closure parameters of roms_*.in
    INTEGER(c_int) :: gls_mixing, gls_stability, gls_n2s2_horavg, gls_ri_splines
    REAL(c_double) :: gls_p, gls_m, gls_n, gls_cmu0, gls_c1, gls_c2, gls_c3m, gls_c3p, gls_sigk, gls_sigp, gls_Kmin, gls_Pmin
    REAL(c_double) :: Akk_bak, Akp_bak, Zos
    !  WET_DRY: switch and the critical depth Dcrit (m); point_sources: LuvSrc.or.LwSrc (refused when non-zero)
    INTEGER(c_int) :: wet_dry, point_sources
    REAL(c_double) :: Dcrit
    !  ATM_PRESS: Pair (mb) in the baroclinic pressure gradient
    INTEGER(c_int) :: atm_press, press_compensate
    !  TS_MIX_STABILITY: 3/4 t(nrhs) + 1/4 t(nstp) in the lateral tracer mixing
    INTEGER(c_int) :: ts_mix_stability
    !  TS_MIX_MIN_STRAT: the slope scale of the isopycnal operator bounded by a minimum stratification
    INTEGER(c_int) :: ts_mix_min_strat
  END TYPE roms_params_t

  !  mirrors `roms_halo_msg_t` of include/roms_hip.h (host relay of the halo exchange)
  TYPE, BIND(C), PUBLIC :: roms_halo_msg_t
    INTEGER(c_int) :: peer, tag
    INTEGER(c_long) :: count
    TYPE(c_ptr) :: buf
  END TYPE roms_halo_msg_t

  !  field identifiers = enum roms_field_id (order of include/roms_fields.def)
  INTEGER(c_int), PARAMETER, PUBLIC :: FID_zeta=0, FID_ubar=1, FID_vbar=2, FID_rzeta=3,    &
 &    FID_rubar=4, FID_rvbar=5, FID_u=6, FID_v=7, FID_t=8, FID_ru=9, FID_rv=10, FID_W=11,   &
 &    FID_rho=12, FID_pden=13, FID_h=14, FID_f=15, FID_fomn=16, FID_pm=17, FID_pn=18,       &
 &    FID_om_r=19, FID_on_r=20, FID_om_u=21, FID_on_u=22, FID_om_v=23, FID_on_v=24,         &
 &    FID_om_p=25, FID_on_p=26, FID_omn=27, FID_pmon_r=28, FID_pnom_r=29, FID_pmon_p=30,    &
 &    FID_pnom_p=31, FID_pmon_u=32, FID_pnom_u=33, FID_pmon_v=34, FID_pnom_v=35,            &
 &    FID_dmde=36, FID_dndx=37, FID_Hz=38, FID_Huon=39, FID_Hvom=40, FID_z_r=41,            &
 &    FID_z_w=42, FID_DU_avg1=43, FID_DU_avg2=44, FID_DV_avg1=45, FID_DV_avg2=46,           &
 &    FID_Zt_avg1=47, FID_rufrc=48, FID_rvfrc=49, FID_rhoA=50, FID_rhoS=51, FID_Akv=52,     &
 &    FID_Akt=53, FID_ghats=54, FID_bvf=55, FID_alpha=56, FID_beta=57, FID_visc2_p=58,      &
 &    FID_visc2_r=59, FID_diff2=60, FID_sustr=61, FID_svstr=62, FID_bustr=63,               &
 &    FID_bvstr=64, FID_srflx=65, FID_stflx=66, FID_btflx=67,                          &
 &    FID_rdrag2=68, FID_stflux=69, FID_btflux=70, FID_Uwind=71, FID_Vwind=72, FID_Tair=73,  &
 &    FID_Pair=74, FID_Hair=75, FID_rain=76, FID_cloud=77, FID_lrflx=78, FID_lhflx=79,      &
 &    FID_shflx=80, FID_evap=81, FID_hsbl=82, FID_rdrag=83, FID_wvel=84, FID_lonr=85, FID_latr=86,   &
 &    FID_rmask=87, FID_umask=88, FID_vmask=89, FID_pmask=90, FID_zeta_bry=91, FID_ubar_bry=92,   &
 &    FID_vbar_bry=93, FID_u_bry=94, FID_v_bry=95, FID_t_bry=96,                                  &
 &    FID_visc4_p=97, FID_visc4_r=98, FID_diff4=99, FID_ZoBot=100,                             &
 &    FID_tke=101, FID_gls=102, FID_Lscale=103, FID_Akk=104, FID_Akp=105,                      &
 &    FID_pmask_wet=106, FID_rmask_wet=107, FID_umask_wet=108, FID_vmask_wet=109, FID_rmask_wet_avg=110,   &
 &    FID_pmask_full=111, FID_rmask_full=112, FID_umask_full=113, FID_vmask_full=114

  INTERFACE
    INTEGER(c_int) FUNCTION roms_hip_init (rank, ntileI, ntileJ, device_id, uid)            &
 &                 BIND(C, name='roms_hip_init')
      IMPORT :: c_int, c_ptr
      INTEGER(c_int), VALUE :: rank, ntileI, ntileJ, device_id
      TYPE(c_ptr), VALUE :: uid
    END FUNCTION roms_hip_init
    INTEGER(c_int) FUNCTION roms_hip_finalize () BIND(C, name='roms_hip_finalize')
      IMPORT :: c_int
    END FUNCTION roms_hip_finalize
    INTEGER(c_int) FUNCTION roms_hip_get_unique_id (out128) BIND(C, name='roms_hip_get_unique_id')
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: out128
    END FUNCTION roms_hip_get_unique_id
    INTEGER(c_int) FUNCTION roms_hip_set_bounds (b) BIND(C, name='roms_hip_set_bounds')
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: b          ! c_loc of a TYPE(roms_bounds_t), TARGET variable
    END FUNCTION roms_hip_set_bounds
    INTEGER(c_int) FUNCTION roms_hip_set_params (p) BIND(C, name='roms_hip_set_params')
      IMPORT :: c_int, c_ptr
      TYPE(c_ptr), VALUE :: p
    END FUNCTION roms_hip_set_params
    INTEGER(c_int) FUNCTION roms_hip_register_field (id, host_ptr, n) BIND(C, name='roms_hip_register_field')
      IMPORT :: c_int, c_ptr, c_long
      INTEGER(c_int), VALUE :: id
      TYPE(c_ptr), VALUE :: host_ptr   ! c_loc(OCEAN(ng)%t(LBi,LBj,1,1,1)) etc.
      INTEGER(c_long), VALUE :: n
    END FUNCTION roms_hip_register_field
    INTEGER(c_int) FUNCTION roms_hip_sync_to_device (id) BIND(C, name='roms_hip_sync_to_device')
      IMPORT :: c_int
      INTEGER(c_int), VALUE :: id
    END FUNCTION roms_hip_sync_to_device
    INTEGER(c_int) FUNCTION roms_hip_sync_to_host (id) BIND(C, name='roms_hip_sync_to_host')
      IMPORT :: c_int
      INTEGER(c_int), VALUE :: id
    END FUNCTION roms_hip_sync_to_host
    INTEGER(c_int) FUNCTION roms_hip_sync_all_to_device () BIND(C, name='roms_hip_sync_all_to_device')
      IMPORT :: c_int
    END FUNCTION roms_hip_sync_all_to_device
    !  debugging aid: 0 when no kernel stored outside a device array (guard bands intact)
    INTEGER(c_int) FUNCTION roms_hip_check_guards () BIND(C, name='roms_hip_check_guards')
      IMPORT :: c_int
    END FUNCTION roms_hip_check_guards
    ! LOOP_2D with its RCCL exchanges as one hipGraph (several tiles; every rank switches it on before the first loop)
    INTEGER(c_int) FUNCTION roms_hip_graph_exchanges (on) BIND(C, name='roms_hip_graph_exchanges')
      IMPORT :: c_int
      INTEGER(c_int), VALUE :: on
    END FUNCTION roms_hip_graph_exchanges
    INTEGER(c_int) FUNCTION roms_hip_graph_exchanges_state () BIND(C, name='roms_hip_graph_exchanges_state')
      IMPORT :: c_int
    END FUNCTION roms_hip_graph_exchanges_state
    INTEGER(c_int) FUNCTION roms_hip_sync_all_to_host () BIND(C, name='roms_hip_sync_all_to_host')
      IMPORT :: c_int
    END FUNCTION roms_hip_sync_all_to_host
    FUNCTION roms_hip_last_error () BIND(C, name='roms_hip_last_error') RESULT(msg)
      IMPORT :: c_ptr
      TYPE(c_ptr) :: msg
    END FUNCTION roms_hip_last_error
    !  fn: INTEGER(c_int) FUNCTION (user, nsend, send, nrecv, recv) BIND(C) with TYPE(roms_halo_msg_t) arrays, e.g.
    !  the MPI_Isend / MPI_Irecv / MPI_Waitall of mp_exchange2d (see roms_hip_demo_mpi.F90)
    INTEGER(c_int) FUNCTION roms_hip_set_halo_relay (fn, user) BIND(C, name='roms_hip_set_halo_relay')
      IMPORT :: c_int, c_funptr, c_ptr
      TYPE(c_funptr), VALUE :: fn
      TYPE(c_ptr), VALUE :: user
    END FUNCTION roms_hip_set_halo_relay
    INTEGER(c_int) FUNCTION roms_hip_step2d_loop (s, indx1) BIND(C, name='roms_hip_step2d_loop')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(inout) :: s
      INTEGER(c_int), INTENT(inout) :: indx1
    END FUNCTION roms_hip_step2d_loop
  END INTERFACE

  !  one interface per kernel entry, all `int f(const roms_step_idx_t*)`
  ABSTRACT INTERFACE
    INTEGER(c_int) FUNCTION roms_hip_entry (s) BIND(C)
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION roms_hip_entry
  END INTERFACE
  INTERFACE
    INTEGER(c_int) FUNCTION roms_hip_set_massflux (s) BIND(C, name='roms_hip_set_massflux')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_rho_eos (s) BIND(C, name='roms_hip_rho_eos')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_omega (s) BIND(C, name='roms_hip_omega')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_set_zeta (s) BIND(C, name='roms_hip_set_zeta')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_set_depth (s) BIND(C, name='roms_hip_set_depth')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_ini_zeta (s) BIND(C, name='roms_hip_ini_zeta')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_ini_fields (s) BIND(C, name='roms_hip_ini_fields')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_rhs3d (s) BIND(C, name='roms_hip_rhs3d')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_step2d (s) BIND(C, name='roms_hip_step2d')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_step3d_uv (s) BIND(C, name='roms_hip_step3d_uv')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_step3d_t (s) BIND(C, name='roms_hip_step3d_t')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_bulk_flux (s) BIND(C, name='roms_hip_bulk_flux')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_set_vbc (s) BIND(C, name='roms_hip_set_vbc')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_lmd_vmix (s) BIND(C, name='roms_hip_lmd_vmix')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    !  asynchronous snapshot for output / wrt_his / wrt_rst: ids(n) = FID_* of the fields wanted
    INTEGER(c_int) FUNCTION roms_hip_snapshot_begin (ids, n) BIND(C, name='roms_hip_snapshot_begin')
      IMPORT :: c_int
      INTEGER(c_int), INTENT(in) :: ids(*)
      INTEGER(c_int), VALUE :: n
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_snapshot_end () BIND(C, name='roms_hip_snapshot_end')
      IMPORT :: c_int
    END FUNCTION
    !  ana_srflux, ALBEDO branch: CALL caldate (tdays(ng), yd_dp=yday, h_dp=hour) on the host, then this
    INTEGER(c_int) FUNCTION roms_hip_ana_srflux (yday, hour) BIND(C, name='roms_hip_ana_srflux')
      IMPORT :: c_int, c_double
      REAL(c_double), VALUE :: yday, hour
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_wvelocity (s) BIND(C, name='roms_hip_wvelocity')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    !  WET_DRY: the initial wet/dry masks (wetdry with Linitialize, initial.F:438-466)
    INTEGER(c_int) FUNCTION roms_hip_wetdry (s) BIND(C, name='roms_hip_wetdry')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    !  LuvSrc: SOURCES(ng) of mod_sources.F (Isrc, Jsrc, Dsrc, Qbar, Qsrc, Tsrc) and LtracerSrc(:,ng) as 0 / 1;
    !  after every set_data that changes them
    INTEGER(c_int) FUNCTION roms_hip_set_sources (Nsrc, Isrc, Jsrc, Dsrc, Qbar, Qsrc, Tsrc, LtracerSrc)            &
   &                        BIND(C, name='roms_hip_set_sources')
      IMPORT :: c_int, c_double
      INTEGER(c_int), VALUE :: Nsrc
      INTEGER(c_int), INTENT(in) :: Isrc(*), Jsrc(*), LtracerSrc(*)
      REAL(c_double), INTENT(in) :: Dsrc(*), Qbar(*), Qsrc(*), Tsrc(*)
    END FUNCTION
    !  GLS_MIXING: gls_prestep (main3d.F:567) and gls_corstep (main3d.F:793)
    INTEGER(c_int) FUNCTION roms_hip_gls_prestep (s) BIND(C, name='roms_hip_gls_prestep')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    INTEGER(c_int) FUNCTION roms_hip_gls_corstep (s) BIND(C, name='roms_hip_gls_corstep')
      IMPORT :: c_int, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
    END FUNCTION
    !  tile-local part of diag_tile; out12 = my_volume, my_avgke, my_avgpe, my_maxspeed, my_maxrho,
    !  my_max_C, my_max_Cu, my_max_Cv, my_max_Cw, my_max_Ci, my_max_Cj, my_max_Ck (diag.F:190-290)
    INTEGER(c_int) FUNCTION roms_hip_diag (s, out12) BIND(C, name='roms_hip_diag')
      IMPORT :: c_int, c_double, roms_step_idx_t
      TYPE(roms_step_idx_t), INTENT(in) :: s
      REAL(c_double), INTENT(out) :: out12(12)
    END FUNCTION
  END INTERFACE

  PUBLIC :: roms_hip_init, roms_hip_finalize, roms_hip_get_unique_id
  PUBLIC :: roms_hip_set_bounds, roms_hip_set_params, roms_hip_register_field
  PUBLIC :: roms_hip_sync_to_device, roms_hip_sync_to_host, roms_hip_set_halo_relay
  PUBLIC :: roms_hip_sync_all_to_device, roms_hip_sync_all_to_host, roms_hip_last_error
  PUBLIC :: roms_hip_set_massflux, roms_hip_rho_eos, roms_hip_omega, roms_hip_set_zeta
  PUBLIC :: roms_hip_set_depth, roms_hip_rhs3d, roms_hip_step2d, roms_hip_step2d_loop
  PUBLIC :: roms_hip_step3d_uv, roms_hip_step3d_t, roms_hip_bulk_flux, roms_hip_set_vbc, roms_hip_lmd_vmix
  PUBLIC :: roms_hip_ana_srflux, roms_hip_wvelocity, roms_hip_diag, roms_hip_snapshot_begin, roms_hip_snapshot_end
  PUBLIC :: roms_hip_ini_zeta, roms_hip_ini_fields, roms_hip_gls_prestep, roms_hip_gls_corstep, roms_hip_wetdry
  PUBLIC :: roms_hip_set_sources
  PUBLIC :: roms_hip_entry, roms_hip_make_idx, roms_hip_status

CONTAINS

  !  Fill the index block from mod_stepping / mod_scalars values
  !  (main3d.F:189-191 and :597-662 keep them current).
  FUNCTION roms_hip_make_idx (iic, ntfirst, nstp, nnew, nrhs, kstp, krhs, knew, iif, predictor) RESULT(s)
    INTEGER, INTENT(in) :: iic, ntfirst, nstp, nnew, nrhs, kstp, krhs, knew, iif
    LOGICAL, INTENT(in) :: predictor
    TYPE(roms_step_idx_t) :: s
    s%iic = iic;   s%ntfirst = ntfirst
    s%nstp = nstp; s%nnew = nnew; s%nrhs = nrhs
    s%kstp = kstp; s%krhs = krhs; s%knew = knew
    s%iif = iif
    s%predictor_2d_step = MERGE(1_c_int, 0_c_int, predictor)
  END FUNCTION roms_hip_make_idx

  !  Map the library's return code onto ROMS' exit_flag convention
  !  (mod_scalars.F:523-532): 0 = NoError, 2 = communication (as
  !  mp_exchange.F:551), 8 = fatal algorithm result.  The caller then tests
  !  FoundError(exit_flag, NoError, __LINE__, MyFile) as main3d.F:178 does.
  SUBROUTINE roms_hip_status (rc, exit_flag)
    INTEGER(c_int), INTENT(in) :: rc
    INTEGER, INTENT(inout) :: exit_flag
    IF (rc == 0) RETURN
    IF (rc == 2) THEN
      exit_flag = 2
    ELSE
      exit_flag = 8
    END IF
  END SUBROUTINE roms_hip_status

END MODULE roms_hip_mod
