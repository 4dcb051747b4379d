/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-thread CPU restatement of the reference's nonlinear 3-D
 * time-stepping kernels (the .F files of ROMS/Nonlinear), written loop-for-loop after the
 * reference so that it can serve as the parity checker of the HIP path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (libroms_hip.so) never links or calls it.
 *
 * Pinning status: see oracle/README.md.  The routines whose reference source
 * compiles stand-alone with flang are checked against that build
 * (oracle/_ref); step2d, pre_step3d, rhs3d, step3d_uv, step3d_t and omega
 * cannot be built from the reference here (mod_sources -> mod_netcdf ->
 * netCDF-Fortran is absent) and are "parity unpinned" apart from shared
 * sub-algorithms and conservation properties.
 *
 * Index macros reproduce the Fortran subscripts, so A(i,j,k) below reads
 * exactly like the reference.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "roms_hip.h"

/* Every kernel starts with ORACLE_PROLOGUE: unpack bounds into locals named
 * as in ROMS/Include/set_bounds.h. */
#define ORACLE_PROLOGUE                                                        \
  const int Lm = b->Lm, Mm = b->Mm, N = b->N, NT = b->NT, NAT = b->NAT;        \
  const int LBi = b->LBi, UBi = b->UBi, LBj = b->LBj, UBj = b->UBj;            \
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;    \
  const int IstrR = b->IstrR, IendR = b->IendR, JstrR = b->JstrR, JendR = b->JendR; \
  const int IstrT = b->IstrT, IendT = b->IendT, JstrT = b->JstrT, JendT = b->JendT; \
  const int IstrP = b->IstrP, IendP = b->IendP, JstrP = b->JstrP, JendP = b->JendP; \
  const int IstrU = b->IstrU, JstrV = b->JstrV;                                \
  const int Istrm3 = b->Istrm3, Istrm2 = b->Istrm2, Istrm1 = b->Istrm1;        \
  const int IstrUm2 = b->IstrUm2, IstrUm1 = b->IstrUm1;                        \
  const int Iendp1 = b->Iendp1, Iendp2 = b->Iendp2, Iendp2i = b->Iendp2i, Iendp3 = b->Iendp3; \
  const int Jstrm3 = b->Jstrm3, Jstrm2 = b->Jstrm2, Jstrm1 = b->Jstrm1;        \
  const int JstrVm2 = b->JstrVm2, JstrVm1 = b->JstrVm1;                        \
  const int Jendp1 = b->Jendp1, Jendp2 = b->Jendp2, Jendp2i = b->Jendp2i, Jendp3 = b->Jendp3; \
  const int IminS = Istr - 3, ImaxS = Iend + 3, JminS = Jstr - 3, JmaxS = Jend + 3; \
  const int EWperiodic = b->EWperiodic, NSperiodic = b->NSperiodic;            \
  const int west_edge = b->west_edge, east_edge = b->east_edge;                \
  const int south_edge = b->south_edge, north_edge = b->north_edge;            \
  const long ni = UBi - LBi + 1, nj = UBj - LBj + 1, nij = ni * nj;            \
  const long n3r = nij * N, n3w = nij * (N + 1);                               \
  const long nis = ImaxS - IminS + 1, njs = JmaxS - JminS + 1;                 \
  (void)Lm; (void)Mm; (void)NT; (void)NAT; (void)UBi; (void)UBj;               \
  (void)IstrR; (void)IendR; (void)JstrR; (void)JendR; (void)IstrT; (void)IendT;\
  (void)JstrT; (void)JendT; (void)IstrP; (void)IendP; (void)JstrP; (void)JendP;\
  (void)IstrU; (void)JstrV; (void)Istrm3; (void)Istrm2; (void)Istrm1;          \
  (void)IstrUm2; (void)IstrUm1; (void)Iendp1; (void)Iendp2; (void)Iendp2i;     \
  (void)Iendp3; (void)Jstrm3; (void)Jstrm2; (void)Jstrm1; (void)JstrVm2;       \
  (void)JstrVm1; (void)Jendp1; (void)Jendp2; (void)Jendp2i; (void)Jendp3;      \
  (void)EWperiodic; (void)NSperiodic; (void)west_edge; (void)east_edge;        \
  (void)south_edge; (void)north_edge; (void)n3r; (void)n3w; (void)nis; (void)njs;\
  (void)ImaxS; (void)JmaxS; (void)IminS; (void)JminS;

#define I2(i,j)    ((long)((i) - LBi) + (long)((j) - LBj) * ni)
#define I3(i,j,k)  (I2(i,j) + (long)((k) - 1) * nij)     /* k = 1..N */
#define I3W(i,j,k) (I2(i,j) + (long)(k) * nij)           /* k = 0..N */

/* ---- module arrays (names as in the reference) ---- */
#define zeta(i,j,n)   F->zeta [I2(i,j) + (long)((n)-1) * nij]
#define ubar(i,j,n)   F->ubar [I2(i,j) + (long)((n)-1) * nij]
#define vbar(i,j,n)   F->vbar [I2(i,j) + (long)((n)-1) * nij]
#define rzeta(i,j,n)  F->rzeta[I2(i,j) + (long)((n)-1) * nij]
#define rubar(i,j,n)  F->rubar[I2(i,j) + (long)((n)-1) * nij]
#define rvbar(i,j,n)  F->rvbar[I2(i,j) + (long)((n)-1) * nij]
#define u(i,j,k,n)    F->u [I3(i,j,k)  + (long)((n)-1) * n3r]
#define v(i,j,k,n)    F->v [I3(i,j,k)  + (long)((n)-1) * n3r]
#define t(i,j,k,n,it) F->t [I3(i,j,k)  + ((long)((n)-1) + 3L * ((it)-1)) * n3r]
#define ru(i,j,k,n)   F->ru[I3W(i,j,k) + (long)((n)-1) * n3w]
#define rv(i,j,k,n)   F->rv[I3W(i,j,k) + (long)((n)-1) * n3w]
#define W(i,j,k)      F->W [I3W(i,j,k)]
#define rho(i,j,k)    F->rho [I3(i,j,k)]
#define pden(i,j,k)   F->pden[I3(i,j,k)]
#define h(i,j)        F->h[I2(i,j)]
#define f(i,j)        F->f[I2(i,j)]
#define fomn(i,j)     F->fomn[I2(i,j)]
#define pm(i,j)       F->pm[I2(i,j)]
#define pn(i,j)       F->pn[I2(i,j)]
#define om_r(i,j)     F->om_r[I2(i,j)]
#define on_r(i,j)     F->on_r[I2(i,j)]
#define om_u(i,j)     F->om_u[I2(i,j)]
#define on_u(i,j)     F->on_u[I2(i,j)]
#define om_v(i,j)     F->om_v[I2(i,j)]
#define on_v(i,j)     F->on_v[I2(i,j)]
#define om_p(i,j)     F->om_p[I2(i,j)]
#define on_p(i,j)     F->on_p[I2(i,j)]
#define omn(i,j)      F->omn[I2(i,j)]
#define pmon_r(i,j)   F->pmon_r[I2(i,j)]
#define pnom_r(i,j)   F->pnom_r[I2(i,j)]
#define pmon_p(i,j)   F->pmon_p[I2(i,j)]
#define pnom_p(i,j)   F->pnom_p[I2(i,j)]
#define pmon_u(i,j)   F->pmon_u[I2(i,j)]
#define pnom_u(i,j)   F->pnom_u[I2(i,j)]
#define pmon_v(i,j)   F->pmon_v[I2(i,j)]
#define pnom_v(i,j)   F->pnom_v[I2(i,j)]
#define dmde(i,j)     F->dmde[I2(i,j)]
#define dndx(i,j)     F->dndx[I2(i,j)]
#define Hz(i,j,k)     F->Hz  [I3(i,j,k)]
#define Huon(i,j,k)   F->Huon[I3(i,j,k)]
#define Hvom(i,j,k)   F->Hvom[I3(i,j,k)]
#define z_r(i,j,k)    F->z_r [I3(i,j,k)]
#define z_w(i,j,k)    F->z_w [I3W(i,j,k)]
#define DU_avg1(i,j)  F->DU_avg1[I2(i,j)]
#define DU_avg2(i,j)  F->DU_avg2[I2(i,j)]
#define DV_avg1(i,j)  F->DV_avg1[I2(i,j)]
#define DV_avg2(i,j)  F->DV_avg2[I2(i,j)]
#define Zt_avg1(i,j)  F->Zt_avg1[I2(i,j)]
#define rufrc(i,j)    F->rufrc[I2(i,j)]
#define rvfrc(i,j)    F->rvfrc[I2(i,j)]
#define rhoA(i,j)     F->rhoA[I2(i,j)]
#define rhoS(i,j)     F->rhoS[I2(i,j)]
#define Akv(i,j,k)    F->Akv[I3W(i,j,k)]
#define Akt(i,j,k,it) F->Akt[I3W(i,j,k) + (long)((it)-1) * n3w]
#define ghats(i,j,k,it) F->ghats[I3W(i,j,k) + (long)((it)-1) * n3w]
#define bvf(i,j,k)    F->bvf[I3W(i,j,k)]
#define alpha(i,j)    F->alpha[I2(i,j)]
#define beta(i,j)     F->beta[I2(i,j)]
#define visc2_p(i,j)  F->visc2_p[I2(i,j)]
#define visc2_r(i,j)  F->visc2_r[I2(i,j)]
#define diff2(i,j,it) F->diff2[I2(i,j) + (long)((it)-1) * nij]
#define sustr(i,j)    F->sustr[I2(i,j)]
#define svstr(i,j)    F->svstr[I2(i,j)]
#define bustr(i,j)    F->bustr[I2(i,j)]
#define bvstr(i,j)    F->bvstr[I2(i,j)]
#define srflx(i,j)    F->srflx[I2(i,j)]
#define stflx(i,j,it) F->stflx[I2(i,j) + (long)((it)-1) * nij]
#define btflx(i,j,it) F->btflx[I2(i,j) + (long)((it)-1) * nij]
#define rdrag2(i,j)   F->rdrag2[I2(i,j)]
#define rdrag(i,j)    F->rdrag[I2(i,j)]
#define wvel(i,j,k)   F->wvel[I3W(i,j,k)]
#define lonr(i,j)     F->lonr[I2(i,j)]
#define latr(i,j)     F->latr[I2(i,j)]
#define stflux(i,j,it) F->stflux[I2(i,j) + (long)((it)-1) * nij]
#define btflux(i,j,it) F->btflux[I2(i,j) + (long)((it)-1) * nij]
#define Uwind(i,j)    F->Uwind[I2(i,j)]
#define Vwind(i,j)    F->Vwind[I2(i,j)]
#define Tair(i,j)     F->Tair[I2(i,j)]
#define Pair(i,j)     F->Pair[I2(i,j)]
#define Hair(i,j)     F->Hair[I2(i,j)]
#define rain(i,j)     F->rain[I2(i,j)]
#define cloud(i,j)    F->cloud[I2(i,j)]
#define lrflx(i,j)    F->lrflx[I2(i,j)]
#define lhflx(i,j)    F->lhflx[I2(i,j)]
#define shflx(i,j)    F->shflx[I2(i,j)]
#define rmask(i,j)    F->rmask[I2(i,j)]
#define umask(i,j)    F->umask[I2(i,j)]
#define vmask(i,j)    F->vmask[I2(i,j)]
#define pmask(i,j)    F->pmask[I2(i,j)]
#define pmask_wet(i,j)     F->pmask_wet[I2(i,j)]
#define rmask_wet(i,j)     F->rmask_wet[I2(i,j)]
#define umask_wet(i,j)     F->umask_wet[I2(i,j)]
#define vmask_wet(i,j)     F->vmask_wet[I2(i,j)]
#define rmask_wet_avg(i,j) F->rmask_wet_avg[I2(i,j)]
#define pmask_full(i,j)    F->pmask_full[I2(i,j)]
#define rmask_full(i,j)    F->rmask_full[I2(i,j)]
#define umask_full(i,j)    F->umask_full[I2(i,j)]
#define vmask_full(i,j)    F->vmask_full[I2(i,j)]
#define zeta_bry(i,j) F->zeta_bry[I2(i,j)]
#define ubar_bry(i,j) F->ubar_bry[I2(i,j)]
#define vbar_bry(i,j) F->vbar_bry[I2(i,j)]
#define u_bry(i,j,k)  F->u_bry[I3(i,j,k)]
#define v_bry(i,j,k)  F->v_bry[I3(i,j,k)]
#define t_bry(i,j,k,it) F->t_bry[I3(i,j,k) + (long)((it)-1) * n3r]

/* private (automatic) work arrays of the _tile routines */
#define WS2(i,j)   ((long)((i) - IminS) + (long)((j) - JminS) * nis)   /* (IminS:ImaxS,JminS:JmaxS) */
#define WSK(i,k)   ((long)((i) - IminS) + (long)(k) * nis)             /* (IminS:ImaxS,0:N)         */
#define WS3(i,j,k) (WS2(i,j) + (long)((k) - 1) * nis * njs)            /* (..,..,N)                 */
static inline double *walloc(long n) { return (double *)calloc((size_t)n, sizeof(double)); }

#define MAX(a,b) ((a) > (b) ? (a) : (b))
#define MIN(a,b) ((a) < (b) ? (a) : (b))

/* grid-point type codes for the periodic exchange (exchange_2d.F/_3d.F) */
enum { GT_R = 0, GT_U, GT_V, GT_P };

#define OARGS const roms_bounds_t *b, const roms_params_t *p, const roms_step_idx_t *s, roms_fields_t *F

/* ---- helpers (oracle_base.c) ---- */
typedef void (*o_exchange_hook_t)(double *A, int nk, int gtype);
void oracle_set_exchange_hook(o_exchange_hook_t fn);
void o_exchange2d(const roms_bounds_t *b, int gtype, double *A);
void o_exchange3d(const roms_bounds_t *b, int gtype, int nk, double *A);
int  o_check_lbc(const roms_bounds_t *b, const roms_params_t *p);
int  o_lbc(const roms_params_t *p, int side, int var);
void o_bc_generic(const roms_bounds_t *b, const roms_params_t *p, const roms_fields_t *F, int gtype, int lbv, double *A,
                  int nk);        /* bc_2d.F / bc_3d.F */
void o_zetabc(OARGS, int kout);
void o_u2dbc(OARGS, int kout);
void o_v2dbc(OARGS, int kout);
void o_u3dbc(OARGS, int nout);
void o_v3dbc(OARGS, int nout);
void o_t3dbc(OARGS, int nout, int itrc);
void o_bc_w3d(const roms_bounds_t *b, double *A);

/* ---- kernels ---- */
int oracle_set_massflux(OARGS);
int oracle_omega(OARGS);
int oracle_set_zeta(OARGS);
int oracle_set_depth(OARGS);
int oracle_rho_eos(OARGS);
int oracle_pre_step3d(OARGS);
int oracle_prsgrd(OARGS);
int oracle_t3dmix2(OARGS);
int oracle_t3dmix4(OARGS);       /* oracle_mix4.c */
int oracle_uv3dmix4(OARGS);
int oracle_t3dmix2_iso(OARGS);
int oracle_rhs3d_tile(OARGS);
int oracle_uv3dmix2(OARGS);
int oracle_uv3dmix2_geo(OARGS);      /* uv3dmix2_geo.h (uv_vis2 = 2: MIX_GEO_UV) */
int oracle_rhs3d(OARGS);
int oracle_step2d(OARGS);
int oracle_step3d_uv(OARGS);
int oracle_step3d_t(OARGS);
int oracle_gls_prestep(OARGS);     /* oracle_gls.c: gls_prestep.F:66 */
int oracle_gls_corstep(OARGS);     /* gls_corstep.F:101 */
int oracle_wetdry(OARGS);          /* oracle_wetdry.c: wetdry.F:17 with Linitialize (wetdry_ini_tile, :395) */
void o_wetdry(OARGS);              /* wetdry_tile, wetdry.F:93 (called by step2d) */
/* the factor of the barotropic / boundary wet-dry rule, e.g. step2d_LF_AM3.h:2124-2126: 1 on a face between two wet
 * cells (mask 2), 0 between two dry ones (0), and on a one-sided face (mask +-1) 1 only for flow out of the wet cell */
static inline double o_wet_factor(double mask_wet, double vel)
{
  const double cff5 = fabs(fabs(mask_wet) - 1.0);
  const double cff6 = 0.5 + copysign(0.5, vel) * mask_wet;
  return 0.5 * mask_wet * cff5 + cff6 * (1.0 - cff5);
}
/* TS_MIX_STABILITY: the tracer difference a - b of the lateral mixing operators, a2 - b2 the same difference of the
 * t(nstp) level (t3dmix2_s.h:212-218 and the like sites of t3dmix2_geo/_iso.h and the first operator of t3dmix4_*.h) */
static inline double o_tdiff(int stab, double a, double b, double a2, double b2)
{
  return stab ? 0.75 * (a - b) + 0.25 * (a2 - b2) : a - b;
}
/* point sources (oracle_sources.c): the table SOURCES(ng), process-wide */
typedef struct { int n, N, NT, given; int ltr[ROMS_MAXNT]; int *I, *J; double *D, *Qbar, *Qsrc, *Tsrc; } o_src_t;
extern o_src_t o_src;
int oracle_set_sources(int Nsrc, const int *Isrc, const int *Jsrc, const double *Dsrc, const double *Qbar,
                       const double *Qsrc, const double *Tsrc, const int *LtracerSrc, int N, int NT);
int o_src_check(const roms_params_t *p);
void o_src_ubar(OARGS, int knew);
void o_src_uv(OARGS, int nnew);
void o_src_tflux(OARGS, int itrc, int k, double *FX_, double *FE_, int wide, int pre);
void o_src_masks(OARGS);
void o_src_zeta(OARGS, int knew);
void o_src_omega(OARGS, int j);
void o_src_wtracer(OARGS, int itrc, int mpdata, int j, double *Ta_, const double *oHz_);
int oracle_ini_zeta(OARGS);        /* ini_fields.F:836 */
int oracle_ini_fields(OARGS);      /* ini_fields.F:106 */
int oracle_step2d_loop(const roms_bounds_t *b, const roms_params_t *p, roms_step_idx_t *s,
                       roms_fields_t *F, int *indx1);
int roms_abi_sizeof(int which);
#endif
