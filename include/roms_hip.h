/*
 * roms_hip.h -- C ABI of the MI355X-native ROMS nonlinear 3-D time-stepping
 * hot path (libroms_hip.so).
 *
 * The reference has no FFI; its seam is the Fortran module-procedure name
 * `CALL X(ng, tile)` used by ROMS/Nonlinear/main3d.F:307-814 and
 * ROMS/Nonlinear/initial.F:337-571.  Each entry point below replaces the body
 * of one such procedure (the file:line it replaces is cited on the
 * declaration).  The ISO_C_BINDING interface block a maintainer adds on the
 * Fortran side is in roms_trunk_mgh_amd/fortran/roms_hip_mod.F90 and
 * INTEGRATION.md.
 *
 * Conventions
 *   - plain C: ints, doubles, pointers; no C++/torch types.
 *   - every entry returns int: 0 = ROMS NoError; non-zero = error, mapped by
 *     the Fortran shim to exit_flag 8 (algorithm) or 2 (communication), the
 *     codes of ROMS/Modules/mod_scalars.F:523-532.  The library never aborts.
 *   - host arrays stay owned by the caller (Fortran `allocate`); the library
 *     owns device mirrors and device scratch (the `_tile` routines' automatic
 *     IminS:ImaxS work arrays).
 *   - array layout = the reference's: column-major, i fastest, common
 *     horizontal extents LBi:UBi,LBj:UBj for every field of a tile.
 */
#ifndef ROMS_HIP_H
#define ROMS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Field identifiers and shapes                                        */
/* ------------------------------------------------------------------ */
enum roms_kind {
  K_2D = 0, K_2D_T2, K_2D_T3, K_2D_NT,
  K_3DR, K_3DW, K_3DR_T2, K_3DW_T2, K_3DW_NAT, K_4DT, K_3DR_NT, K_3DW_T3
};

enum roms_field_id {
#define ROMS_FIELD(name, kind, owner) FID_##name,
#include "roms_fields.def"
#undef ROMS_FIELD
  FID_COUNT
};

/* All module arrays of one tile (host pointers in the oracle, device pointers
 * inside the library). */
typedef struct roms_fields {
#define ROMS_FIELD(name, kind, owner) double *name;
#include "roms_fields.def"
#undef ROMS_FIELD
} roms_fields_t;

/* ------------------------------------------------------------------ */
/* Tile bounds: every integer of ROMS/Include/set_bounds.h:20-79 and   */
/* ROMS/Include/tile.h:21-45, plus DOMAIN(ng)%*_Edge(tile)             */
/* (ROMS/Modules/mod_param.F:286-323) and the grid sizes.             */
/* ------------------------------------------------------------------ */
typedef struct roms_bounds {
  int Lm, Mm, N, NT, NAT;
  int ntileI, ntileJ, tile, Itile, Jtile;
  int NghostPoints, EWperiodic, NSperiodic;
  int west_edge, east_edge, south_edge, north_edge;
  int LBi, UBi, LBj, UBj;
  int Istr, Iend, Jstr, Jend;
  int IstrB, IendB, IstrM, IstrP, IendP, IstrR, IendR, IstrT, IendT, IstrU;
  int JstrB, JendB, JstrM, JstrP, JendP, JstrR, JendR, JstrT, JendT, JstrV;
  int Istrm3, Istrm2, Istrm1, IstrUm2, IstrUm1;
  int Iendp1, Iendp2, Iendp2i, Iendp3;
  int Jstrm3, Jstrm2, Jstrm1, JstrVm2, JstrVm1;
  int Jendp1, Jendp2, Jendp2i, Jendp3;
} roms_bounds_t;

/* ------------------------------------------------------------------ */
/* Run-time parameters (mod_scalars.F, mod_param.F)                    */
/* ------------------------------------------------------------------ */
#define ROMS_MAXN   64     /* max vertical levels held in the param block   */
#define ROMS_MAXNT  16     /* max tracers                                   */
#define ROMS_MAXFAST 256   /* max 2*ndtfast                                 */

/* Tracer advection scheme codes = the logical records of T_ADV
 * (mod_param.F:382-394). */
enum roms_adv {
  ADV_C2 = 0, ADV_C4, ADV_A4, ADV_U3, ADV_SU3, ADV_SPLINES, ADV_MPDATA, ADV_HSIMT
};
/* Pressure-gradient algorithm (the CPP choice of ROMS/Nonlinear/prsgrd.F:16-26): DJ_GRADPS = prsgrd32.h (the three
 * application headers of BASELINE.json define it), none of the options = prsgrd31.h (standard density Jacobian, the
 * reference's default), WJ_GRADP = prsgrd31.h with the weighted Jacobian of Song (1998). */
enum roms_pgf { PGF_DJ_GRADPS = 0, PGF_STANDARD = 1, PGF_WJ_GRADP = 2, PGF_PJ_GRADP = 3 /* prsgrd40.h: finite-volume pressure Jacobian */ };
/* Lateral boundary condition codes supported on this path (the logical records of T_LBC, mod_param.F:348-363).
 * A periodic direction (E-W) has LBC_PERIODIC on both of its sides; a physical edge (any of the four) takes, per
 * variable (roms_params_t.lbc): closed, gradient, clamped, radiation (implicit upstream, zetabc.F:108 /
 * u2dbc_im.F:135 / v2dbc_im.F:136 / u3dbc_im.F:131 / v3dbc_im.F:97 / t3dbc_im.F:128 and the other edges' blocks; no
 * nudging, RADIATION_2D off) for every variable; Chapman implicit for zeta (zetabc.F:193); Flather for ubar and
 * vbar (u2dbc_im.F:214, v2dbc_im.F:216 for the normal component,
 * [radiation + nudging towards the boundary data, "RadNud", for every variable: e.g. t3dbc_im.F:138-152, :183-188] the Chapman-type rule of u2dbc_im.F:912 /
 * v2dbc_im.F:886 for the tangential one); then the corner rule (zetabc.F:699).  Anything else is refused by every
 * entry that applies conditions; N-S periodic grids are refused. */
enum roms_lbc {
  LBC_PERIODIC = 0, LBC_CLOSED = 1, LBC_GRADIENT = 2, LBC_CLAMPED = 3, LBC_CHAPMAN_IMPLICIT = 4, LBC_FLATHER = 5,
  LBC_RADIATION = 6, LBC_RADIATION_NUDGING = 7,
  LBC_CHAPMAN_EXPLICIT = 8,          /* "Che", free surface only: zetabc.F:175-190 */
  LBC_SHCHEPETKIN = 9,               /* "Shc", ubar / vbar only: u2dbc_im.F:288-362, v2dbc_im.F:290-364 (bry_val = the boundary data) */
  LBC_REDUCED = 10                   /* "Red", ubar / vbar only: reduced physics (pressure gradient, Coriolis, surface and
                                      * bottom stress), u2dbc_im.F:392-432, v2dbc_im.F:394-436 */
};
/* rows of roms_params_t.lbc = the state variables of LBC(:, isFsur / isUbar / isVbar / isUvel / isVvel / isTvar, ng) */
enum roms_lbc_var { LBV_ZETA = 0, LBV_UBAR, LBV_VBAR, LBV_U, LBV_V, LBV_T, LBV_COUNT };
enum roms_lbc_side { LBS_WEST = 0, LBS_EAST, LBS_SOUTH, LBS_NORTH };

typedef struct roms_params {
  double dt, dtfast;                 /* mod_scalars.F dt(ng), dtfast(ng)     */
  double g, rho0;                    /* mod_scalars.F:431-441                */
  double gamma2;                     /* slipperiness, roms_*.in GAMMA2       */
  double lambda;                     /* implicit weight, mod_scalars.F:724   */
  int    ndtfast, nfast;             /* set_weights.F:3-244                  */
  double weight1[ROMS_MAXFAST];      /* weight(1,1:2*ndtfast,ng)             */
  double weight2[ROMS_MAXFAST];      /* weight(2,1:2*ndtfast,ng)             */
  int    Vtransform;                 /* set_depth.F:82                       */
  int    limit_bstress;              /* LIMIT_BSTRESS: the bottom stress may only slow the bottom velocity down to zero
                                      * within a step, set_vbc.F:533-540, :562-567 (sits in the alignment gap before hc) */
  double hc;
  double sc_r[ROMS_MAXN + 1], Cs_r[ROMS_MAXN + 1];   /* index 1..N           */
  double sc_w[ROMS_MAXN + 1], Cs_w[ROMS_MAXN + 1];   /* index 0..N           */
  int    Hadv[ROMS_MAXNT], Vadv[ROMS_MAXNT];         /* enum roms_adv        */
  int    lbc_west, lbc_east, lbc_south, lbc_north;   /* enum roms_lbc        */
  /* equation of state */
  int    nonlin_eos;                 /* 1 = NONLIN_EOS (rho_eos.F:111)       */
  int    eminusp;                    /* EMINUSP: bulk_flux also sets evap = LHeat / Hlv and the surface salt flux
                                      * stflux(isalt) = (evap - rain) / rhow, bulk_flux.F:883-899 (alignment gap before R0) */
  double R0, T0, S0, Tcoef, Scoef;   /* linear EOS (rho_eos.F:576)           */
  /* CPP-derived option switches of the application header */
  int    uv_adv, uv_cor, uv_vis2, curvgrid, var_rho_2d;   /* uv_vis2: 0 = no UV_VIS2, 1 = MIX_S_UV (uv3dmix2_s.h:114),
                                                          * 2 = MIX_GEO_UV (uv3dmix2_geo.h:116) */
  int    ts_dif2, mix_geo_ts, mix_s_ts, salinity, lmd_nonlocal, solar_source;
  int    splines_vdiff, splines_vvisc;
  double Akt_bak[ROMS_MAXNT], Akv_bak;
  /* Jerlov water type constants of lmd_swfrac.F:6 (mod_scalars.F:1502-1512), uniform WTYPE */
  double swfrac_mu1, swfrac_mu2, swfrac_r1;
  /* per-step physics between the hot kernels */
  int    uv_drag;                    /* bottom stress law of set_vbc.F: 1 = UV_LDRAG, 2 = UV_QDRAG, 3 = UV_LOGDRAG */
  int    mpdata_fast;                /* library switch (no reference counterpart): 1 = the anti-diffusive
                                      * velocities of mpdata_adiff use refined reciprocals instead of IEEE
                                      * divisions (results within the 1e-10 relative-RMS bound of the exact
                                      * kernel, not bit-identical); 0 = exact */
  double blk_ZQ, blk_ZT, blk_ZW;     /* measurement heights of bulk_flux.F (roms_*.in BLK_ZQ/ZT/ZW) */
  int    masking;                    /* 1 = the application defines MASKING: rmask/umask/vmask/pmask are applied
                                      * where the reference applies them (e.g. step2d_LF_AM3.h:778, step3d_t.F:603) */
  int    pgf;                        /* enum roms_pgf: the pressure-gradient algorithm prsgrd.F:16-26 selects */
  /* lbc[side][variable] (enum roms_lbc_side, roms_lbc_var; every tracer shares LBV_T): 0 = take the side's
   * lbc_west / lbc_east / lbc_south / lbc_north above, otherwise an enum roms_lbc code */
  int    lbc[4][LBV_COUNT];
  /* LBC_RADIATION_NUDGING ("RadNud"): nudging coefficients (1/s) of the boundary point towards the boundary data,
   * per side and variable -- FSobc_out/in (zeta), M2obc_out/in (ubar, vbar), M3obc_out/in (u, v), Tobc_out/in (t; one
   * value for all tracers) of mod_scalars.F:1336-1365, i.e. 1/(xNUDG*86400) and OBCFAC times that */
  double obc_out[4][LBV_COUNT], obc_in[4][LBV_COUNT];
  /* biharmonic lateral mixing (TS_DIF4, UV_VIS4 of the application header; coefficients diff4, visc4_r, visc4_p =
   * the square roots the reference stores, inp_par.F:986, read_phypar.F:6905).  Tracers: along s-surfaces (mix_s_ts,
   * t3dmix4_s.h:23) or geopotentials (mix_geo_ts, t3dmix4_geo.h:23); momentum: along s-surfaces (uv3dmix4_s.h:23)
   * and the 2-D operator of step2d_LF_AM3.h:1494-1740.  TS_DIF2 and TS_DIF4 (UV_VIS2 and UV_VIS4) may both be set. */
  int    ts_dif4, uv_vis4;
  /* MIX_ISO_TS: tracer mixing along isopycnals (t3dmix2_iso.h:23, t3dmix4_iso.h:23) -- reads pden of rho_eos; takes
   * precedence over mix_geo_ts / mix_s_ts.  The default slope treatment (none of TS_MIX_MAX_SLOPE, TS_MIX_MIN_STRAT). */
  int    mix_iso_ts;
  /* RADIATION_2D: the radiation conditions (LBC_RADIATION, LBC_RADIATION_NUDGING; 2-D and 3-D variables) include the
   * tangential phase speed Ce (e.g. zetabc.F:141-147, t3dbc_im.F:151-165) */
  int    radiation_2d;
  /* UV_LOGDRAG (uv_drag = 3): limits of the drag coefficient of the logarithmic bottom layer, roms_*.in Cdb_min /
   * Cdb_max (mod_scalars.F:747-748); the roughness length is the field ZoBot */
  double Cdb_min, Cdb_max;
  /* GLS_MIXING: the generic length-scale closure of Umlauf and Burchard (2003) as gls_prestep.F / gls_corstep.F build
   * it (main3d.F:564-567, :790-793); the parameter sets of roms_*.in (GLS_P ... GLS_SIGP: k-kl = Mellor-Yamada 2.5,
   * k-epsilon, k-omega, gen) select the closure.  gls_stability: enum roms_gls_stab (the CPP choice GALPERIN (none),
   * KANTHA_CLAYSON, CANUTO_A, CANUTO_B); gls_n2s2_horavg = N2S2_HORAVG; gls_ri_splines = RI_SPLINES (the shear from
   * parabolic splines); the third-order upstream advection of tke / gls (neither K_C2ADVECTION nor K_C4ADVECTION).
   * CRAIG_BANNER, CHARNOK, ZOS_HSIG, TKE_WAVEDISS are not built.  Akk_bak, Akp_bak: background diffusivities of
   * tke and gls; Zos: surface roughness (m), mod_scalars.F.
   * gls_mixing = 2: MY25_MIXING instead -- the same two entries are then my25_prestep (my25_prestep.F:23, the text of
   * gls_prestep.F) and my25_corstep (my25_corstep.F:27; tke = q2, gls = q2l; Galperin et al. stability functions, Sm
   * of Kantha and Clayson with gls_stability = GLS_KANTHA_CLAYSON; N2S2_HORAVG, RI_SPLINES as above).  Of the
   * parameters only gls_Kmin, gls_Pmin (initial values) and Akk_bak are read; Akp and ZoBot are not used. */
  int    gls_mixing, gls_stability, gls_n2s2_horavg, gls_ri_splines;
  double gls_p, gls_m, gls_n, gls_cmu0, gls_c1, gls_c2, gls_c3m, gls_c3p, gls_sigk, gls_sigp, gls_Kmin, gls_Pmin;
  double Akk_bak, Akp_bak, Zos;
  /* WET_DRY (wetdry.F, step2d_LF_AM3.h:729-755, :863-866, :2123-2160 ...): cells whose total depth falls to Dcrit
   * (roms_*.in DCRIT, m) are masked out of the barotropic and baroclinic stepping.  The wet/dry masks are fields
   * (pmask_wet ... vmask_full); roms_hip_wetdry initialises them (initial.F:438-466), every step2d call updates
   * them.  With wet_dry the barotropic step takes the general launch sequence (flux, free surface, masks,
   * momentum). */
  int    wet_dry;
  /* Point sources / sinks of mod_sources.F (rivers): bit 0 = LuvSrc(ng) (transport through u- / v-faces, Dsrc = 0 / 1),
   * bit 1 = LwSrc(ng) (volume influx at cell centres, Dsrc = 2).  With LuvSrc the table comes from roms_hip_set_sources;
   * until it has been handed over every entry refuses to run (error text "point sources"), so that a river application
   * cannot lose its sources silently.  LwSrc is not built: refused.  0 = the application has none. */
  int    point_sources;
  double Dcrit;
  /* ATM_PRESS (prsgrd32.h:229-232, :264-266; prsgrd31.h:196-198, :213-215, :294-296; prsgrd40.h:187-196): the
   * atmospheric surface pressure Pair (mb, FID_Pair) in the baroclinic pressure gradient (inverse barometer).
   * press_compensate = PRESS_COMPENSATE (with ATM_PRESS): the same term in the Flather value of the normal barotropic
   * velocity (u2dbc_im.F:264-272, :612-620; v2dbc_im.F:266, :615). */
  int    atm_press, press_compensate;
  /* TS_MIX_STABILITY (t3dmix2_s.h:212-218, t3dmix2_geo.h:236-239 / :268 / :301, t3dmix2_iso.h:239 / :271 / :321, the
   * first operator of t3dmix4_s.h:262 / :308, t3dmix4_geo.h:279 / :311 / :345, t3dmix4_iso.h:287 / :319 / :369): every
   * tracer difference of the lateral mixing operator is 3/4 of t(nrhs)'s plus 1/4 of t(nstp)'s. */
  int    ts_mix_stability;
  /* TS_MIX_MIN_STRAT (t3dmix2_iso.h:313-316, t3dmix4_iso.h:361-364, :679-682; with MIX_ISO_TS): the density difference
   * that scales the isopycnal slopes is at least strat_min = 0.1 kg/m3 per metre times the layer distance, in the
   * place of the constant eps = 0.5. */
  int    ts_mix_min_strat;
} roms_params_t;
enum roms_gls_stab { GLS_GALPERIN = 0, GLS_KANTHA_CLAYSON = 1, GLS_CANUTO_A = 2, GLS_CANUTO_B = 3 };

/* Time-level indices = mod_stepping.F (nstp,nnew,nrhs,kstp,krhs,knew) and
 * mod_scalars.F (iic, iif, ntfirst, PREDICTOR_2D_STEP); see
 * main3d.F:189-191 and main3d.F:597-662.  All 1-based as in Fortran. */
typedef struct roms_step_idx {
  int iic, ntfirst;
  int nstp, nnew, nrhs;
  int kstp, krhs, knew;
  int iif, predictor_2d_step;
} roms_step_idx_t;

/* ------------------------------------------------------------------ */
/* Library life cycle                                                  */
/* ------------------------------------------------------------------ */
/* Replaces nothing; called once after ROMS_initialize (nl_roms.h:60-231).
 * nccl_unique_id: 128-byte ncclUniqueId shared by all ranks, or NULL (one tile, or a multi-tile run whose
 * halos go through roms_hip_set_halo_relay).  One tile WITH an id = loopback: the periodic wrap of the tile
 * is sent through RCCL to the tile itself and every kernel takes its multi-tile branch (same results; lets a
 * one-GPU box execute the RCCL transport). */
int roms_hip_init(int rank, int ntileI, int ntileJ, int device_id,
                  const void *nccl_unique_id);
int roms_hip_finalize(void);
/* 128-byte id for roms_hip_init, generated on rank 0 (ncclGetUniqueId). */
int roms_hip_get_unique_id(void *out128);

/* BOUNDS(ng)%...(tile): get_bounds.F:738 (get_tile), :1009 (var_bounds). */
int roms_hip_set_bounds(const roms_bounds_t *b);
int roms_hip_set_params(const roms_params_t *p);

/* Register the host address (c_loc) of one module array; allocates its
 * device mirror.  n_doubles is checked against the shape implied by kind. */
int roms_hip_register_field(int field_id, double *host_ptr, long n_doubles);
/* Host <-> device copies of one field (whole array). */
int roms_hip_sync_to_device(int field_id);
int roms_hip_sync_to_host(int field_id);
int roms_hip_sync_all_to_device(void);
int roms_hip_sync_all_to_host(void);
/* Asynchronous snapshot for the unchanged output / wrt_his / wrt_rst (ROMS/Nonlinear/output.F:123-208):
 * begin() copies the n listed fields aside on the device (ordered with the kernels already issued) and starts
 * their transfer to the registered host arrays on a second stream, then returns; the step loop may go on and
 * overwrite the fields.  end() waits for the transfer: the host arrays then hold the fields as they were at
 * begin().  One snapshot in flight at a time; the host arrays must not be read or written in between. */
int roms_hip_snapshot_begin(const int *field_ids, int n);
int roms_hip_snapshot_end(void);
/* Device address of a field mirror (for zero-copy consumers); NULL if none.  The grid-metric arrays are taken to
 * be constant between uploads (roms_hip_row_metrics_state below): a consumer that WRITES one of them through this
 * pointer must upload or re-register the field afterwards. */
double *roms_hip_device_ptr(int field_id);
int roms_hip_device_synchronize(void);
const char *roms_hip_last_error(void);

/* ------------------------------------------------------------------ */
/* Hot-path entry points -- one per reference module procedure         */
/* ------------------------------------------------------------------ */
/* set_massflux(ng,tile,model)      ROMS/Nonlinear/set_massflux.F:28  */
int roms_hip_set_massflux(const roms_step_idx_t *s);
/* rho_eos(ng,tile,model)           ROMS/Nonlinear/rho_eos.F:47       */
int roms_hip_rho_eos(const roms_step_idx_t *s);
/* omega(ng,tile,model)             ROMS/Nonlinear/omega.F:28         */
int roms_hip_omega(const roms_step_idx_t *s);
/* set_zeta(ng,tile)                ROMS/Nonlinear/set_zeta.F:25      */
int roms_hip_set_zeta(const roms_step_idx_t *s);
/* set_depth(ng,tile,model)         ROMS/Nonlinear/set_depth.F:33     */
int roms_hip_set_depth(const roms_step_idx_t *s);
/* Informational: how the barotropic kernel reads the fifteen grid-metric arrays (pm, pn, on_u, om_v, fomn, dndx,
 * dmde, pmon_r, pnom_r, pmon_p, pnom_p, om_r, on_r, om_p, on_p).  0 = not examined since their last upload;
 * 1 = all of them are independent of i on this tile (checked bit for bit on the device: a zonally uniform grid) and
 * the kernel takes them from a per-row table; 2 = they are not, and it reads the arrays; 3 = as 1, and the resting
 * depth h and the viscosity coefficients visc2_r, visc2_p are independent of i as well (flat or zonally uniform
 * bathymetry) and come from the table too.  Same results in every case. */
int roms_hip_row_metrics_state(void);
/* ini_zeta(ng,tile,model)          ROMS/Nonlinear/ini_fields.F:780
 * ini_fields(ng,tile,model)        ROMS/Nonlinear/ini_fields.F:27
 * The first-step initialisation of main3d.F:269-283 (ini_zeta, set_depth, ini_fields, in this order): other
 * time levels loaded from the initial state, MASKING multiplies, lateral boundary conditions, ubar/vbar = the
 * vertical means of u/v, Zt_avg1 = the initial free surface.  Uses s->kstp, knew, nstp, nnew. */
int roms_hip_ini_zeta(const roms_step_idx_t *s);
int roms_hip_ini_fields(const roms_step_idx_t *s);
/* rhs3d(ng,tile) -> pre_step3d, prsgrd, t3dmix2, t3dmix4, rhs3d_tile, uv3dmix2, uv3dmix4 (each mixing call
 * under its switch ts_dif2 / ts_dif4 / uv_vis2 / uv_vis4)
 *                                  ROMS/Nonlinear/rhs3d.F:25         */
int roms_hip_rhs3d(const roms_step_idx_t *s);
/* the pieces of rhs3d, exported for per-kernel parity tests */
int roms_hip_pre_step3d(const roms_step_idx_t *s);  /* pre_step3d.F:39   */
int roms_hip_prsgrd(const roms_step_idx_t *s);      /* prsgrd32.h:40     */
int roms_hip_t3dmix2(const roms_step_idx_t *s);     /* t3dmix2_geo.h:23  */
int roms_hip_rhs3d_tile(const roms_step_idx_t *s);  /* rhs3d.F:174       */
int roms_hip_uv3dmix2(const roms_step_idx_t *s);    /* uv3dmix2_s.h:42; with uv_vis2 = 2 uv3dmix2_geo.h:42 */
int roms_hip_t3dmix4(const roms_step_idx_t *s);     /* t3dmix4_s.h:23, t3dmix4_geo.h:23 (TS_DIF4) */
int roms_hip_uv3dmix4(const roms_step_idx_t *s);    /* uv3dmix4_s.h:43 (UV_VIS4; three ghost points, inp_par.F:268) */
/* step2d(ng,tile)                  ROMS/Nonlinear/step2d_LF_AM3.h:18 */
int roms_hip_step2d(const roms_step_idx_t *s);
/* step3d_uv(ng,tile)               ROMS/Nonlinear/step3d_uv.F:27     */
int roms_hip_step3d_uv(const roms_step_idx_t *s);
/* step3d_t(ng,tile)                ROMS/Nonlinear/step3d_t.F:40      */
int roms_hip_step3d_t(const roms_step_idx_t *s);

/* Per-step physics between the hot kernels, main3d.F:388-430:
 * bulk_flux(ng,tile)               ROMS/Nonlinear/bulk_flux.F:46     */
int roms_hip_bulk_flux(const roms_step_idx_t *s);
/* set_vbc(ng,tile)                 ROMS/Nonlinear/set_vbc.F:34       */
int roms_hip_set_vbc(const roms_step_idx_t *s);
/* lmd_vmix(ng,tile) = lmd_vmix_tile + lmd_skpp + lmd_finish   ROMS/Nonlinear/lmd_vmix.F:37 */
int roms_hip_lmd_vmix(const roms_step_idx_t *s);
/* ana_srflux(ng,tile,model)        ROMS/Functionals/ana_srflux.h:2, the ALBEDO branch (:120-150) the BENCHMARK
 * application uses: shortwave radiation from the zenith angle at (lonr, latr) for the day of the year and the
 * hour that caldate (ROMS/Utility/dateclock.F:73) returns for tdays(ng) -- the host passes those two numbers --
 * with the cloud and water-vapour corrections from cloud, Tair, Hair.  Writes srflx. */
int roms_hip_ana_srflux(double yday, double hour);
/* gls_prestep(ng,tile)             ROMS/Nonlinear/gls_prestep.F:23   (main3d.F:567, after rhs3d)
 * gls_corstep(ng,tile)             ROMS/Nonlinear/gls_corstep.F:27   (main3d.F:793, after omega and before step3d_t)
 * with tkebc_tile (tkebc_im.F:50: closed and gradient edges).  GLS_MIXING applications only: these two replace
 * lmd_vmix as the source of Akv / Akt.
 * my25_prestep(ng,tile)            ROMS/Nonlinear/my25_prestep.F:23  (main3d.F:565)
 * my25_corstep(ng,tile)            ROMS/Nonlinear/my25_corstep.F:27  (main3d.F:791)
 * MY25_MIXING applications bind the same two entries with roms_params_t.gls_mixing = 2. */
int roms_hip_gls_prestep(const roms_step_idx_t *s);
int roms_hip_gls_corstep(const roms_step_idx_t *s);
/* wetdry(ng,tile,Tindex,.TRUE.)    ROMS/Nonlinear/wetdry.F:17 -> wetdry_ini_tile (:395): the initial wet/dry masks
 * from zeta, ubar, vbar of time level Tindex = s->kstp (initial.F:438-466; WET_DRY applications).  The per-call
 * update wetdry_tile (:93) runs inside roms_hip_step2d. */
int roms_hip_wetdry(const roms_step_idx_t *s);
/* The source table SOURCES(ng) of mod_sources.F:56-80 with LuvSrc (roms_params_t.point_sources bit 0): Isrc, Jsrc (grid
 * indices of the u- or v-face), Dsrc (0.0 = u-face, 1.0 = v-face), Qbar(Nsrc) (m3/s), Qsrc(Nsrc,N) = Qbar * Qshape as
 * set_data.F:136-143 leaves it, Tsrc(Nsrc,N,NT) and LtracerSrc(NT); Fortran element order.  Call it after every
 * set_data that changes them (the arrays are copied; a few kB).  The entries that consume it: step2d
 * (step2d_LF_AM3.h:2484-2502), step3d_uv (step3d_uv.F:971-995), pre_step3d (pre_step3d.F:530-553), step3d_t
 * (step3d_t.F:734-799), wetdry (wetdry.F:307-320, :511-524).  A source with Dsrc = 2 (LwSrc) is refused. */
int roms_hip_set_sources(int Nsrc, const int *Isrc, const int *Jsrc, const double *Dsrc, const double *Qbar,
                         const double *Qsrc, const double *Tsrc, const int *LtracerSrc);
/* wvelocity(ng,tile,nstp)          ROMS/Nonlinear/wvelocity.F:27     (main3d.F:475; writes wvel) */
int roms_hip_wvelocity(const roms_step_idx_t *s);
/* diag(ng,tile)                    ROMS/Nonlinear/diag.F:31          (main3d.F:314), the tile-local part
 * diag.F:190-290 on time level nstp: out12 = { my_volume, my_avgke, my_avgpe, my_maxspeed, my_maxrho,
 * my_max_C, my_max_Cu, my_max_Cv, my_max_Cw, my_max_Ci, my_max_Cj, my_max_Ck } (host memory; the call
 * returns when they are there).  The reduction over tiles -- mp_reduce SUM/SUM/SUM/MAX/MAX and
 * mp_reduce2 MAXLOC, diag.F:398-420 -- and the printing stay with the caller. */
int roms_hip_diag(const roms_step_idx_t *s, double *out12);

/* The whole barotropic loop LOOP_2D of main3d.F:592-700 in one call
 * (predictor/corrector sequencing done inside: 2*nfast+1 launches queued on the library's
 * stream without returning to the host).  indx1 is mod_stepping's indx1(ng), updated on return. */
int roms_hip_step2d_loop(roms_step_idx_t *s, int *indx1);

/* Several tiles over RCCL: replay LOOP_2D as ONE hipGraph that contains the compute launches, the pack / unpack
 * launches and the ncclSend / ncclRecv groups of its 2*nfast+1 calls (RCCL enqueues its kernels on the capturing
 * stream).  Every rank of the communicator must switch it on before its first roms_hip_step2d_loop, since all of
 * them then capture once and replay the same sequence.  Off by default: on this stack the replay shortens the loop's
 * device time (BENCHMARK1 tile, loopback: 1.85 -> 1.65 ms) but lengthens the step's wall time (3.05 -> 3.64 ms).  If
 * the stack refuses the capture the loop keeps running eagerly.  state: 0 = eager, 1 = graph in use, -1 = capture was
 * tried and refused.  (the reference's loop is step2d_LF_AM3.h:509-590 per call.) */
int roms_hip_graph_exchanges(int on);
int roms_hip_graph_exchanges_state(void);

/* mp_exchange2d/3d/4d (ROMS/Utility/mp_exchange.F:290/1413/2753) together
 * with the periodic exchange_*_tile (exchange_2d.F:229, exchange_3d.F:259) on
 * whole registered fields; level = 1-based trailing index (time level or
 * tracer) or 0 for all.  Exported for tests. */
int roms_hip_exchange(int field_id, int level);

/* Second halo transport: a host relay.  When roms_hip_init got no RCCL id (NULL) on a
 * multi-tile run, every exchange packs its messages on the device (one per neighbour tile: W, E, S, N
 * and the four diagonal ones, as far as they exist), copies them to pinned host memory and calls `fn`,
 * which must deliver send[m].buf (count doubles) to rank send[m].peer with message tag send[m].tag and
 * fill recv[m].buf from rank recv[m].peer, tag recv[m].tag -- e.g. with the MPI_Isend / MPI_Irecv /
 * MPI_Waitall the reference's mp_exchange2d (ROMS/Utility/mp_exchange.F:290-560) already uses.  Two
 * messages between the same pair of ranks differ in their tag.  Return 0 on success.
 * The RCCL transport is the fast one; the relay exists for hosts that own the
 * interconnect and for rehearsing N tiles on fewer GPUs. */
typedef struct { int peer; int tag; long count; double *buf; } roms_halo_msg_t;
typedef int (*roms_halo_relay_fn)(void *user, int nsend, const roms_halo_msg_t *send,
                                  int nrecv, const roms_halo_msg_t *recv);
int roms_hip_set_halo_relay(roms_halo_relay_fn fn, void *user);
/* Host-only (no GPU): the messages tile `rank` exchanges per halo update for the bounds *b:
 * out = nsend, nrecv, then (peer, tag, i0, wi, j0, wj) per message, sends first; at most 8 + 8 messages
 * (98 ints).  Exported for the CPU tests, which play a whole 4x2 exchange with it. */
int roms_hip_halo_plan(const roms_bounds_t *b, int rank, int *out);

/* Timing helper: average device milliseconds of the last call of each entry
 * measured with hipEvents on the library's stream (bench.py roofline). */
int roms_hip_timing_enable(int on);
double roms_hip_timing_last_ms(const char *entry);

/* Profiling aid (no reference counterpart): one streaming copy of n_doubles
 * from 3-D scratch array 0 to scratch array 1 in the library's access pattern
 * (one double per lane, i-fastest).  A known byte count -- 8*n read, 8*n
 * written -- against which rocprofv3's FETCH_SIZE / WRITE_SIZE counters are
 * calibrated for this pattern.  n_doubles <= nij*(N+1). */
int roms_hip_calib_stream(long n_doubles);

/* Debugging aid (no reference counterpart): every device mirror and scratch array carries a guard band of
 * four rows on either side, filled with one NaN bit pattern.  Returns 0 when every band is intact, an
 * error naming the array otherwise (a kernel stored outside LBi:UBi,LBj:UBj at the first or last plane).
 * Reads in the band are harmless by construction and deliver NaN. */
int roms_hip_check_guards(void);

#ifdef __cplusplus
}
#endif
#endif /* ROMS_HIP_H */
