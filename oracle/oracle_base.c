/*
 * oracle_base.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Periodic exchanges, closed lateral boundary conditions and the small glue
 * kernels (set_massflux, omega, set_zeta, set_depth).
 */
#include "oracle.h"

int roms_abi_sizeof(int which)
{
  switch (which) {
  case 0: return (int)sizeof(roms_bounds_t);
  case 1: return (int)sizeof(roms_params_t);
  case 2: return (int)sizeof(roms_step_idx_t);
  case 3: return (int)sizeof(roms_fields_t);
  case 4: return (int)FID_COUNT;
  }
  return -1;
}

/* Effective boundary-condition code (enum roms_lbc) of variable v (enum roms_lbc_var) on side sd. */
int o_lbc(const roms_params_t *p, int sd, int v)
{
  if (p->lbc[sd][v]) return p->lbc[sd][v];
  return sd == LBS_WEST ? p->lbc_west : sd == LBS_EAST ? p->lbc_east : sd == LBS_SOUTH ? p->lbc_south : p->lbc_north;
}

/* Which conditions are restated (oracle_bc.c), per variable: closed, gradient, clamped, radiation (all six),
 * Chapman implicit (zeta), Flather (ubar, vbar: the normal component; the reference applies a Chapman-type
 * condition to the tangential component of a Flather edge).  A periodic direction must be periodic on both of its
 * sides; a physical edge takes one of the conditions above. */
int o_check_lbc(const roms_bounds_t *b, const roms_params_t *p)
{
  for (int sd = LBS_WEST; sd <= LBS_NORTH; sd++) {
    const int periodic = (sd <= LBS_EAST) ? b->EWperiodic : b->NSperiodic;
    for (int v = 0; v < LBV_COUNT; v++) {
      const int c = o_lbc(p, sd, v);
      if (periodic) {
        if (c != LBC_PERIODIC) return 1;
        continue;
      }
      int ok = c == LBC_CLOSED || c == LBC_GRADIENT || c == LBC_CLAMPED || c == LBC_RADIATION || c == LBC_RADIATION_NUDGING;
      if (v == LBV_ZETA) ok = ok || c == LBC_CHAPMAN_IMPLICIT || c == LBC_CHAPMAN_EXPLICIT;
      if (v == LBV_VBAR || v == LBV_UBAR) ok = ok || c == LBC_FLATHER || c == LBC_SHCHEPETKIN || c == LBC_REDUCED;
      if (!ok) return 1;
    }
  }
  return 0;
}

/* exchange_{r,u,v,p}2d_tile -- ROMS/Nonlinear/exchange_2d.F:43/229/416/603.
 * Periodic ghost copy, only when the direction is not partitioned. */
/* Multi-tile runs (CPU tests of the N>1 protocol): the test harness installs a
 * hook that performs mp_exchange2d/3d (mp_exchange.F:290/1413) between ranks. */
static o_exchange_hook_t g_hook = 0;
void oracle_set_exchange_hook(o_exchange_hook_t fn) { g_hook = fn; }

static void o_periodic2d(const roms_bounds_t *b, int gtype, double *A);

void o_exchange2d(const roms_bounds_t *b, int gtype, double *A)
{
  o_periodic2d(b, gtype, A);
  if (g_hook && b->ntileI * b->ntileJ > 1) g_hook(A, 1, gtype);
}

static void o_periodic2d(const roms_bounds_t *b, int gtype, double *A)
{
  const int Lm = b->Lm, LBi = b->LBi, LBj = b->LBj;
  const long ni = b->UBi - b->LBi + 1;
  if (!(b->EWperiodic && b->ntileI == 1)) return;
  int Jmin, Jmax;
  if (b->NSperiodic) { Jmin = b->Jstr; Jmax = b->Jend; }
  else {
    Jmin = (gtype == GT_R || gtype == GT_U) ? b->JstrR : b->Jstr;
    Jmax = b->JendR;
  }
#define A2(i,j) A[(long)((i) - LBi) + (long)((j) - LBj) * ni]
  if (b->west_edge) {
    for (int j = Jmin; j <= Jmax; j++) {
      A2(Lm + 1, j) = A2(1, j);
      A2(Lm + 2, j) = A2(2, j);
    }
    if (b->NghostPoints == 3)
      for (int j = Jmin; j <= Jmax; j++) A2(Lm + 3, j) = A2(3, j);
  }
  if (b->east_edge) {
    for (int j = Jmin; j <= Jmax; j++) {
      A2(-2, j) = A2(Lm - 2, j);
      A2(-1, j) = A2(Lm - 1, j);
      A2(0, j) = A2(Lm, j);
    }
  }
#undef A2
}

/* exchange_{r,u,v,w}3d_tile -- ROMS/Nonlinear/exchange_3d.F:259/471/683/896 */
void o_exchange3d(const roms_bounds_t *b, int gtype, int nk, double *A)
{
  const long nij = (long)(b->UBi - b->LBi + 1) * (b->UBj - b->LBj + 1);
  for (int k = 0; k < nk; k++) o_periodic2d(b, gtype, A + (long)k * nij);
  if (g_hook && b->ntileI * b->ntileJ > 1) g_hook(A, nk, gtype);
}

/* bc_w3d_tile (zero-gradient on every physical edge, corners, then the periodic wrap / tile exchange) --
 * ROMS/Nonlinear/bc_3d.F:588-725 */
void o_bc_w3d(const roms_bounds_t *b, double *A)
{
  const int LBi = b->LBi, LBj = b->LBj, N = b->N;
  const long ni = b->UBi - b->LBi + 1, nij = ni * (b->UBj - b->LBj + 1);
#define A3(i,j,k) A[(long)((i) - LBi) + (long)((j) - LBj) * ni + (long)(k) * nij]
  for (int k = 0; k <= N; k++) {
    if (!b->EWperiodic) {
      if (b->east_edge) for (int j = b->Jstr; j <= b->Jend; j++) A3(b->Iend + 1, j, k) = A3(b->Iend, j, k);
      if (b->west_edge) for (int j = b->Jstr; j <= b->Jend; j++) A3(b->Istr - 1, j, k) = A3(b->Istr, j, k);
    }
    if (!b->NSperiodic) {
      if (b->north_edge) for (int i = b->Istr; i <= b->Iend; i++) A3(i, b->Jend + 1, k) = A3(i, b->Jend, k);
      if (b->south_edge) for (int i = b->Istr; i <= b->Iend; i++) A3(i, b->Jstr - 1, k) = A3(i, b->Jstr, k);
    }
    if (!b->EWperiodic && !b->NSperiodic) {
      const int iw = b->Istr - 1, ie = b->Iend + 1, js = b->Jstr - 1, jn = b->Jend + 1;
      if (b->south_edge && b->west_edge) A3(iw, js, k) = 0.5 * (A3(iw + 1, js, k) + A3(iw, js + 1, k));
      if (b->south_edge && b->east_edge) A3(ie, js, k) = 0.5 * (A3(ie - 1, js, k) + A3(ie, js + 1, k));
      if (b->north_edge && b->west_edge) A3(iw, jn, k) = 0.5 * (A3(iw, jn - 1, k) + A3(iw + 1, jn, k));
      if (b->north_edge && b->east_edge) A3(ie, jn, k) = 0.5 * (A3(ie, jn - 1, k) + A3(ie - 1, jn, k));
    }
  }
#undef A3
  o_exchange3d(b, GT_R, N + 1, A);
}

/* set_massflux_tile -- ROMS/Nonlinear/set_massflux.F:73-188 */
int oracle_set_massflux(OARGS)
{
  ORACLE_PROLOGUE
  (void)p;
  const int nrhs = s->nrhs;
  for (int k = 1; k <= N; k++) {
    for (int j = JstrT; j <= JendT; j++)
      for (int i = IstrP; i <= IendT; i++)
        Huon(i, j, k) = 0.5 * (Hz(i, j, k) + Hz(i - 1, j, k)) * u(i, j, k, nrhs) * on_u(i, j);
    for (int j = JstrP; j <= JendT; j++)
      for (int i = IstrT; i <= IendT; i++)
        Hvom(i, j, k) = 0.5 * (Hz(i, j, k) + Hz(i, j - 1, k)) * v(i, j, k, nrhs) * om_v(i, j);
  }
  o_exchange3d(b, GT_U, N, F->Huon);
  o_exchange3d(b, GT_V, N, F->Hvom);
  return 0;
}

/* omega_tile -- ROMS/Nonlinear/omega.F:73-229 */
int oracle_omega(OARGS)
{
  ORACLE_PROLOGUE
  if (o_src_check(p)) return 8;
  double *wrk = walloc(nis);
  for (int j = Jstr; j <= Jend; j++) {
    for (int i = Istr; i <= Iend; i++) W(i, j, 0) = 0.0;
    for (int k = 1; k <= N; k++)
      for (int i = Istr; i <= Iend; i++)
        W(i, j, k) = W(i, j, k - 1) -
                     (Huon(i + 1, j, k) - Huon(i, j, k) +
                      Hvom(i, j + 1, k) - Hvom(i, j, k));
    o_src_omega(b, p, s, F, j);                      /* LwSrc, omega.F:165-190 */
    for (int i = Istr; i <= Iend; i++)
      wrk[i - IminS] = W(i, j, N) / (z_w(i, j, N) - z_w(i, j, 0));
    for (int k = N - 1; k >= 1; k--)
      for (int i = Istr; i <= Iend; i++)
        W(i, j, k) = W(i, j, k) - wrk[i - IminS] * (z_w(i, j, k) - z_w(i, j, 0));
    for (int i = Istr; i <= Iend; i++) W(i, j, N) = 0.0;
  }
  free(wrk);
  o_bc_w3d(b, F->W);
  return 0;
}

/* set_zeta_tile -- ROMS/Nonlinear/set_zeta.F:59-129 */
int oracle_set_zeta(OARGS)
{
  ORACLE_PROLOGUE
  (void)p; (void)s;
  for (int j = JstrR; j <= JendR; j++)
    for (int i = IstrR; i <= IendR; i++) {
      zeta(i, j, 1) = Zt_avg1(i, j);
      zeta(i, j, 2) = Zt_avg1(i, j);
    }
  o_exchange2d(b, GT_R, F->zeta);
  o_exchange2d(b, GT_R, F->zeta + nij);
  return 0;
}

/* set_depth_tile -- ROMS/Nonlinear/set_depth.F:82-300 */
int oracle_set_depth(OARGS)
{
  ORACLE_PROLOGUE
  (void)s;
  const double hc = p->hc;
  if (p->Vtransform == 1) {
    for (int j = JstrT; j <= JendT; j++) {
      for (int i = IstrT; i <= IendT; i++) {
        if (p->wet_dry && h(i, j) == 0.0) h(i, j) = 1.0E-14;      /* WET_DRY, set_depth.F:168-172 / :216-220: h itself is changed */
        z_w(i, j, 0) = -h(i, j);
      }
      for (int k = 1; k <= N; k++) {
        double cff_r = hc * (p->sc_r[k] - p->Cs_r[k]);
        double cff_w = hc * (p->sc_w[k] - p->Cs_w[k]);
        double cff1_r = p->Cs_r[k], cff1_w = p->Cs_w[k];
        for (int i = IstrT; i <= IendT; i++) {
          double hwater = h(i, j);
          double hinv = 1.0 / hwater;
          double z_w0 = cff_w + cff1_w * hwater;
          z_w(i, j, k) = z_w0 + Zt_avg1(i, j) * (1.0 + z_w0 * hinv);
          double z_r0 = cff_r + cff1_r * hwater;
          z_r(i, j, k) = z_r0 + Zt_avg1(i, j) * (1.0 + z_r0 * hinv);
          Hz(i, j, k) = z_w(i, j, k) - z_w(i, j, k - 1);
        }
      }
    }
  } else {
    for (int j = JstrT; j <= JendT; j++) {
      for (int i = IstrT; i <= IendT; i++) {
        if (p->wet_dry && h(i, j) == 0.0) h(i, j) = 1.0E-14;      /* WET_DRY, set_depth.F:168-172 / :216-220: h itself is changed */
        z_w(i, j, 0) = -h(i, j);
      }
      for (int k = 1; k <= N; k++) {
        double cff_r = hc * p->sc_r[k], cff_w = hc * p->sc_w[k];
        double cff1_r = p->Cs_r[k], cff1_w = p->Cs_w[k];
        for (int i = IstrT; i <= IendT; i++) {
          double hwater = h(i, j);
          double hinv = 1.0 / (hc + hwater);
          double cff2_r = (cff_r + cff1_r * hwater) * hinv;
          double cff2_w = (cff_w + cff1_w * hwater) * hinv;
          z_w(i, j, k) = Zt_avg1(i, j) + (Zt_avg1(i, j) + hwater) * cff2_w;
          z_r(i, j, k) = Zt_avg1(i, j) + (Zt_avg1(i, j) + hwater) * cff2_r;
          Hz(i, j, k) = z_w(i, j, k) - z_w(i, j, k - 1);
        }
      }
    }
  }
  o_exchange2d(b, GT_R, F->h);
  o_exchange3d(b, GT_R, N + 1, F->z_w);
  o_exchange3d(b, GT_R, N, F->z_r);
  o_exchange3d(b, GT_R, N, F->Hz);
  return 0;
}
